// C ABI of librt_mi355x.so (see include/rt_mi355x.h).  Host-side state and launch logic only;
// all device code is in rt_kernels.hip.  There is no CPU fallback: every entry point needs a
// working HIP device and reports a HIP failure as an error.

#include "../../include/rt_mi355x.h"
#include "rt_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(const std::string &m) {
    g_err = m;
    return 1;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                            \
    } while (0)

struct RawBuffer {           // device copy of an uploaded reference-layout buffer
    uint32_t *raw = nullptr;  // as uploaded
    size_t raw_bytes = 0;
    uint32_t count = 0;       // elements
    float4 *packed = nullptr; // device records (count + 1 elements); for nodes, triangles, spheres and
                              // boxes a region of the context's record arena, an allocation of its own otherwise
    size_t packed_cap = 0;    // capacity in float4 (allocations of their own)
    bool dirty = true;        // packed is stale
};

// Order of the device node records: order[new] = old, rank[old] = new.  Every node appears exactly
// once (nodes the root does not reach keep their relative order at the end; a child index is followed
// only the first time it is met, so malformed graphs are fine).
// The first `top` records are the ones a workgroup copies into LDS: the breadth-first order from
// node 0 for layouts 0 to 2.  Below them, `layout` chooses:
//   0  breadth-first throughout;
//   1  depth-first (pre-order) within the subtree of each node that did not make it into the top:
//      a subtree's nodes are contiguous;
//   2  treelets: the same, but the depth-first walk emits groups of up to 21 nodes (a node, its
//      children, their children: three levels) before descending, so a ray's next few visits lie
//      within a few KB (a van Emde Boas-like blocking);
//   3  breadth-first below the top, but the top itself is not the first `top` nodes of the breadth-first
//      order: it is grown from the root by always taking the frontier node whose box (the child box
//      its parent holds for it, in scene units: `ext` scales the 16-bit codes) has the largest
//      surface area -- the nodes a random ray is most likely to visit (the default: 0.8 % of a frame
//      of the 265 k-triangle bench scene against the plain breadth-first top, nothing on small scenes).
void node_order(const uint32_t *nodes, uint32_t n, uint32_t top, int layout, const float ext[3],
                std::vector<uint32_t> &order, std::vector<uint32_t> &rank) {
    order.clear();
    order.reserve(n);
    rank.assign(n, 0xFFFFFFFFu);
    if (n == 0) return;
    auto children = [&](uint32_t node, uint32_t out[4]) {
        int k = 0;
        const uint32_t *meta = nodes + (size_t)node * 16 + 12;
        for (int c = 0; c < 4; c++) {
            const uint32_t idx = meta[c] & 0x3FFFFFFFu;
            if ((meta[c] >> 30) == 3u && idx < n && rank[idx] == 0xFFFFFFFFu) out[k++] = idx;
        }
        return k;
    };
    auto emit = [&](uint32_t node) {
        rank[node] = (uint32_t)order.size();
        order.push_back(node);
    };
    emit(0);
    if (layout == 3) {
        // (area, node) of the frontier; the largest area first, ties to the smaller node index
        std::vector<std::pair<double, uint32_t>> heap;
        auto less = [](const std::pair<double, uint32_t> &a, const std::pair<double, uint32_t> &b) {
            return a.first < b.first || (a.first == b.first && a.second > b.second);
        };
        auto push_children = [&](uint32_t node) {
            const uint32_t *rec = nodes + (size_t)node * 16;
            for (int c = 0; c < 4; c++) {
                const uint32_t idx = rec[12 + c] & 0x3FFFFFFFu;
                if ((rec[12 + c] >> 30) != 3u || idx >= n) continue;
                double d[3];
                for (int a = 0; a < 3; a++) {
                    const uint32_t w = rec[4 * a + c];
                    d[a] = ((double)(w >> 16) - (double)(w & 0xFFFFu)) * (double)ext[a];
                    if (!(d[a] > 0.0)) d[a] = 0.0;
                }
                heap.emplace_back(d[0] * d[1] + d[1] * d[2] + d[2] * d[0], idx);
                std::push_heap(heap.begin(), heap.end(), less);
            }
        };
        push_children(0);
        while (!heap.empty() && order.size() < (size_t)top) {
            std::pop_heap(heap.begin(), heap.end(), less);
            const uint32_t node = heap.back().second;
            heap.pop_back();
            if (rank[node] != 0xFFFFFFFFu) continue;  // met through another parent already
            emit(node);
            push_children(node);
        }
        // everything else breadth-first from the root (walking through the nodes already placed)
        std::vector<uint32_t> queue(1, 0u);
        std::vector<uint8_t> seen(n, 0);
        seen[0] = 1;
        for (size_t h = 0; h < queue.size(); h++) {
            const uint32_t *meta = nodes + (size_t)queue[h] * 16 + 12;
            for (int c = 0; c < 4; c++) {
                const uint32_t idx = meta[c] & 0x3FFFFFFFu;
                if ((meta[c] >> 30) != 3u || idx >= n || seen[idx]) continue;
                seen[idx] = 1;
                if (rank[idx] == 0xFFFFFFFFu) emit(idx);
                queue.push_back(idx);
            }
        }
    }
    const size_t bfs_limit = layout == 3 ? 0 : layout == 0 ? (size_t)n : (size_t)top;
    size_t head = 0;
    for (; head < order.size() && order.size() < bfs_limit; head++) {
        uint32_t kids[4];
        const int k = children(order[head], kids);
        for (int c = 0; c < k; c++)
            if (rank[kids[c]] == 0xFFFFFFFFu) emit(kids[c]);  // a child met twice within one node counts once
    }
    if (layout == 1 || layout == 2) {
        // the frontier: emitted nodes whose children have not been looked at yet, in breadth-first order
        std::vector<uint32_t> stack;
        for (size_t i = order.size(); i-- > head;) stack.push_back(order[i]);  // popped in breadth-first order
        // every frontier node is already placed; what follows places its descendants
        std::vector<uint32_t> level, next;
        while (!stack.empty()) {
            const uint32_t node = stack.back();
            stack.pop_back();
            if (layout == 1) {
                uint32_t kids[4];
                const int k = children(node, kids);
                for (int c = 0; c < k; c++)
                    if (rank[kids[c]] == 0xFFFFFFFFu) emit(kids[c]);
                // (children are contiguous right after... their own subtrees follow, first child first)
                for (int c = k - 1; c >= 0; c--) stack.push_back(kids[c]);
            } else {
                // two more levels below `node` form one treelet; its deepest level becomes new roots
                level.assign(1, node);
                for (int depth = 0; depth < 2; depth++) {
                    next.clear();
                    for (uint32_t m : level) {
                        uint32_t kids[4];
                        const int k = children(m, kids);
                        for (int c = 0; c < k; c++)
                            if (rank[kids[c]] == 0xFFFFFFFFu) emit(kids[c]), next.push_back(kids[c]);
                    }
                    level.swap(next);
                }
                for (size_t i = level.size(); i-- > 0;) stack.push_back(level[i]);
            }
        }
    }
    for (uint32_t i = 0; i < n; i++)
        if (rank[i] == 0xFFFFFFFFu) {
            rank[i] = (uint32_t)order.size();
            order.push_back(i);
        }
}

}  // namespace

void rt::set_last_error(const std::string &message) { g_err = message; }

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int W = 1280, H = 720;
    RawBuffer spheres, planes, boxes, mats, nodes, tris;
    uint32_t *node_order = nullptr, *node_rank = nullptr;  // device: breadth-first renumbering of the nodes
    size_t node_perm_cap = 0;
    // Nodes, triangles, spheres and boxes -- every record a traversal step may fetch -- in one
    // allocation, in this order, each region beginning on a 128-byte line (DevScene in rt_internal.h).
    float4 *arena = nullptr;
    size_t arena_cap = 0;            // float4
    rt::LeafRegions leaves = {0, 0, 0, 0, 0, 0};
    std::vector<uint32_t> host_nodes;                       // host copy of the uploaded node words (for the renumbering)
    float smin[3] = {0, 0, 0}, sext[3] = {0, 0, 0};
    float inv_vp[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float eye[3] = {0, 0, 0};
    float4 *sample = nullptr, *acc = nullptr, *sum = nullptr;
    uint32_t *aux = nullptr;
    // Two frames in flight: trace kernels alternate between two streams (and two sample / sum
    // images), so the ramp-down of one frame overlaps the start of the next; everything else --
    // uploads, repacks, the accumulate pass, folds, read-backs -- stays on `stream`, ordered
    // against the trace streams with events.
    hipStream_t tstream[2] = {nullptr, nullptr};
    hipEvent_t ev_ready = nullptr;                // `stream` work the trace streams must observe
    hipEvent_t ev_tdone[2] = {nullptr, nullptr};  // last trace kernel of a trace stream
    hipEvent_t ev_acc[2] = {nullptr, nullptr};    // last accumulate pass that read sample image b
    hipEvent_t ev_tlast[2] = {nullptr, nullptr};  // timed: end of the last call's last launch on trace stream b
    bool tlast_used[2] = {false, false};
    bool main_dirty = true;                       // something was queued on `stream` since ev_ready
    bool need_ready[2] = {false, false};
    bool acc_recorded[2] = {false, false};
    bool tpending[2] = {false, false};            // trace stream has work `stream` does not depend on yet
    float4 *sample2 = nullptr, *sum2 = nullptr;   // images of the odd frames
    float4 *sum2_target = nullptr;                // where sum2 is folded into
    bool sum2_dirty = false;
    int flip = 0;                                 // parity of the next frame
    int last_sample = 0;                          // which sample image holds the newest frame
    bool overlap = true;                          // RT_OVERLAP=0 keeps every kernel on `stream`
    rt::Counters *counters = nullptr;
    uint32_t rendered_frames = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_trace;  // pairs around trace kernels of the last call
    size_t ev_trace_used = 0;
    bool timed = false;
    // scheduling of the frame: 0 = megakernel (default: measured fastest), 1 = wavefront (ray queue
    // in HBM, persistent traversal kernel; RT_PIPELINE=wave), 2 = megakernel that regroups a tile's
    // live rays while they traverse (RT_PIPELINE=regroup)
    int pipeline = 0;
    rt::WaveQueues q = {{nullptr, nullptr}, nullptr, nullptr};
    uint64_t pixel_samples = 0;  // pixel-samples queued since the last rt_reset_stats
    // fast mode (opt-in; rt_set_fast_mode): the library's own tree, built on the host from copies
    // of the uploaded triangles, spheres and boxes
    bool fast = false, fast_dirty = true;
    rt::FastBvh fbvh;
    float4 *fnodes = nullptr;
    size_t fnodes_cap = 0;  // bytes
    void *comm = nullptr;  // RCCL communicator (rt_comm_init)
    int comm_rank = 0, comm_world = 1;
    double *comm_scratch = nullptr;
    // Traversal variant of the megakernel (rt_kernels.hip, trace_body): the unified node / leaf step
    // or leaf steps of their own.  Neither wins everywhere.  A new scene starts on the variant its
    // size suggests; once it has rendered kTuneAfterFrames plain frames (or one call of that many
    // arrives: the timing costs about four frames and wins 6 to 18 % of one), four extra
    // launches that write nothing are timed -- leaf steps, unified (both to get the kernels loaded
    // and the caches warm), then leaf steps and unified again -- and the faster of the last two is
    // kept until the scene or the image size changes.  RT_LEAF_STEPS=0 / 1 fixes it.
    int leaf_steps = 0;
    bool tuned = false, tune_fixed = false;
    float tune_ms[4] = {0.f, 0.f, 0.f, 0.f};
    hipEvent_t ev_tune[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    rt::Counters *tune_counters = nullptr;  // where those launches count their rays
    // what the last timing was done on: a scene re-uploaded at about the same size (an animation)
    // keeps the variant instead of paying four launches per upload
    uint64_t tuned_records = 0, tuned_pixels = 0;
    uint32_t plain_frames = 0;  // plain frames rendered since the scene or the size changed, while undecided
    // Macro tiles in the order of what they cost in the previous frame of the same parity, most
    // expensive first, so that a launch does not end on a few long-running workgroups (RT_LPT=0
    // turns it off).  Any order renders every pixel exactly once; only the launch's tail changes.
    // Used for frames that run alone (RT_OVERLAP=0, or no earlier frame still on the device when
    // the frame is queued, as in an interactive loop that reads every frame back): 2.6 % off a
    // launch there; when frames overlap, the next frame fills the tail anyway and the ordering pass
    // only costs (0.6 %).
    hipEvent_t ev_busy[2] = {nullptr, nullptr};  // end of the last trace kernel queued on trace stream b
    bool busy_recorded[2] = {false, false};
    bool lpt = true;
    int node_layout = 3;  // order of the device node records (node_order): which nodes make up the LDS copy and how the
                          // rest follow; RT_NODE_LAYOUT = 0, 1, 2 are the measured alternatives
    uint32_t *lpt_cost[2] = {nullptr, nullptr}, *lpt_order[2] = {nullptr, nullptr};
    float *pinned = nullptr;  // page-locked host image handed out by rt_map_image
    size_t pinned_bytes = 0;
    uint32_t extend_waves = 8192;
    uint32_t refill_threshold = 16;
    bool wave_bin = false;  // wavefront pipeline: rays regrouped by direction octant between bounces (RT_WAVE_BIN=1)
};

namespace {

int stride_of(int binding) {
    switch (binding) {
        case RT_BINDING_SPHERES:
        case RT_BINDING_PLANES:
        case RT_BINDING_BOXES: return 32;
        case RT_BINDING_MATERIALS: return 12;
        case RT_BINDING_BVH:
        case RT_BINDING_TRIANGLES: return 64;
        default: return 0;
    }
}

RawBuffer *buffer_of(rt_ctx *c, int binding) {
    switch (binding) {
        case RT_BINDING_SPHERES: return &c->spheres;
        case RT_BINDING_PLANES: return &c->planes;
        case RT_BINDING_BOXES: return &c->boxes;
        case RT_BINDING_MATERIALS: return &c->mats;
        case RT_BINDING_BVH: return &c->nodes;
        case RT_BINDING_TRIANGLES: return &c->tris;
        default: return nullptr;
    }
}

int vecs_per_elem(int binding) {
    switch (binding) {
        case RT_BINDING_BVH: return rt::kNodeVec;
        case RT_BINDING_TRIANGLES: return rt::kTriVec;
        case RT_BINDING_MATERIALS: return rt::kMatVec;
        default: return 2;
    }
}

// (Re)allocates the images and the ray queue for width x height.  Everything new is allocated
// before anything old is released: on failure the context keeps its previous size and images.
int alloc_images(rt_ctx *c, int width, int height) {
    const size_t n = (size_t)width * height;
    // ray queue of the wavefront pipeline: one slot per lane of the 16x16 tile grid
    const size_t tiles = (size_t)((width / 8 * 8 + 15) / 16) * (size_t)((height / 4 * 4 + 15) / 16);
    const size_t slots = c->pipeline == 1 ? ((tiles + 7) / 8) * 8 * 256 + 256 : 0;  // only that pipeline has a queue
    void **dst[8] = {(void **)&c->sample, (void **)&c->sample2, (void **)&c->acc,      (void **)&c->sum,
                     (void **)&c->sum2,   (void **)&c->q.rays[0], (void **)&c->q.hits, (void **)&c->q.paths};
    const size_t bytes[8] = {n * sizeof(float4), n * sizeof(float4), n * sizeof(float4), n * sizeof(float4),
                             n * sizeof(float4), slots * 2 * sizeof(float4), slots * sizeof(float2),
                             slots * 2 * sizeof(float4)};
    void *fresh[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 8; i++) {
        if (bytes[i] == 0) continue;
        hipError_t e = hipMalloc(&fresh[i], bytes[i]);
        if (e == hipSuccess && i < 5) e = hipMemsetAsync(fresh[i], 0, bytes[i], c->stream);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(c->stream);
            for (int k = 0; k <= i; k++)
                if (fresh[k]) (void)hipFree(fresh[k]);
            (void)hipGetLastError();  // the failure is reported here, not by a later call
            return fail(std::string("allocating a ") + std::to_string(width) + "x" + std::to_string(height) +
                        " image set: " + hipGetErrorString(e));
        }
    }
    for (int i = 0; i < 8; i++) {
        if (*dst[i]) (void)hipFree(*dst[i]);
        *dst[i] = fresh[i];
    }
    if (c->aux) {
        (void)hipFree(c->aux);
        c->aux = nullptr;
    }
    if (c->pinned) {  // pointers handed out by rt_map_image end here
        (void)hipHostFree(c->pinned);
        c->pinned = nullptr, c->pinned_bytes = 0;
    }
    c->W = width;
    c->H = height;
    // rendered_frames stays: the reference's update_size leaves its counter alone (renderer.py:347-382)
    c->main_dirty = true;
    c->sum2_dirty = false;
    c->acc_recorded[0] = c->acc_recorded[1] = false;
    c->flip = c->last_sample = 0;
    return 0;
}

// one frame on stream `s` with the selected pipeline
void trace_frame(rt_ctx *c, const rt::TraceParams &p, int mode, bool aux, bool stats, hipStream_t s) {
    if (c->fast)
        rt::launch_trace_fast(p, mode, aux, stats, s);
    else if (c->pipeline == 0)
        rt::launch_trace(p, mode, aux, stats, c->leaf_steps != 0, s);
    else if (c->pipeline == 2)
        rt::launch_trace_regroup(p, mode, aux, stats, s);
    else
        rt::launch_wavefront_frame(p, c->q, mode, aux, stats, c->extend_waves, c->refill_threshold, c->wave_bin, s);
}

// Makes `stream` depend on everything queued on the trace streams and folds the odd frames' sum
// image into its target.  After this, work queued on `stream` sees every frame rendered so far.
int join(rt_ctx *c) {
    for (int b = 0; b < 2; b++) {
        if (!c->tpending[b]) continue;
        HIP_TRY(hipEventRecord(c->ev_tdone[b], c->tstream[b]));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tdone[b], 0));
        c->tpending[b] = false;
    }
    if (c->sum2_dirty) {
        rt::launch_fold_sum(c->sum2_target, c->sum2, (size_t)c->W * c->H, c->stream);
        HIP_TRY(hipGetLastError());
        c->sum2_dirty = false;
        c->main_dirty = true;
    }
    return 0;
}

// join, then wait on the host
int sync_all(rt_ctx *c) {
    if (int rc = join(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// Stream for the next frame of parity b.  An overlapped frame goes to trace stream b, which first
// waits for whatever `stream` has queued that the frame must observe.
int trace_stream(rt_ctx *c, int b, bool serial, hipStream_t *out) {
    if (serial) {
        if (int rc = join(c)) return rc;
        *out = c->stream;
        return 0;
    }
    if (c->main_dirty) {
        HIP_TRY(hipEventRecord(c->ev_ready, c->stream));
        c->main_dirty = false;
        c->need_ready[0] = c->need_ready[1] = true;
    }
    if (c->need_ready[b]) {
        HIP_TRY(hipStreamWaitEvent(c->tstream[b], c->ev_ready, 0));
        c->need_ready[b] = false;
    }
    *out = c->tstream[b];
    return 0;
}

// brings the device records of one binding with an allocation of its own (planes, materials) up to date
int repack(rt_ctx *c, int binding) {
    RawBuffer *b = buffer_of(c, binding);
    if (!b->dirty) return 0;
    if (int rc = sync_all(c)) return rc;  // frames in flight still read the old records
    c->main_dirty = true;
    size_t need = ((size_t)b->count + 1) * vecs_per_elem(binding) + 8;
    if (need > b->packed_cap) {
        float4 *old = b->packed;
        b->packed = nullptr, b->packed_cap = 0;  // a failure below leaves "no records", not a stale capacity
        if (old) HIP_TRY(hipFree(old));
        HIP_TRY(hipMalloc((void **)&b->packed, need * sizeof(float4)));
        b->packed_cap = need;
    }
    if (binding == RT_BINDING_MATERIALS) rt::launch_repack_mats(b->raw, b->count, b->packed, c->stream);
    else rt::launch_copy_raw32(b->raw, b->count, b->packed, c->stream);
    HIP_TRY(hipGetLastError());
    b->dirty = false;
    return 0;
}

// Brings the record arena up to date: if any of its four bindings changed, the regions are laid out
// anew and all four are repacked -- the child words of the node records name records by position, so
// they depend on where every region begins and on how many records it holds.
int repack_arena(rt_ctx *c) {
    RawBuffer *part[4] = {&c->nodes, &c->tris, &c->spheres, &c->boxes};
    const int vecs[4] = {rt::kNodeVec, rt::kTriVec, rt::kSphereVec, 2};
    if (!(part[0]->dirty || part[1]->dirty || part[2]->dirty || part[3]->dirty)) return 0;
    if (int rc = sync_all(c)) return rc;  // frames in flight still read the old records
    c->main_dirty = true;
    size_t begin[4], total = 0;
    for (int k = 0; k < 4; k++) {
        begin[k] = total;
        // count + 1 records (the closing all-zero one), rounded up to whole 128-byte lines: the traversal
        // fetches records by the line, so the line of the last record must be readable
        total += (((size_t)part[k]->count + 1) * vecs[k] + 7) / 8 * 8;
    }
    if (total / 2 >= (1u << 30)) return fail("the scene's records exceed 32 GB: positions no longer fit a child word");
    if (total > c->arena_cap) {
        float4 *old = c->arena;
        c->arena = nullptr, c->arena_cap = 0;
        for (RawBuffer *b : part) b->packed = nullptr, b->dirty = true;  // a failure below leaves "no records"
        if (old) HIP_TRY(hipFree(old));
        HIP_TRY(hipMalloc((void **)&c->arena, total * sizeof(float4)));
        c->arena_cap = total;
    }
    for (int k = 0; k < 4; k++) part[k]->packed = c->arena + begin[k], part[k]->dirty = true;
    c->leaves = {(uint32_t)(begin[1] / 2), (uint32_t)(begin[2] / 2), (uint32_t)(begin[3] / 2),
                 c->tris.count, c->spheres.count, c->boxes.count};
    {
        RawBuffer *b = &c->nodes;
        std::vector<uint32_t> order, rank;
        node_order(c->host_nodes.data(), b->count, rt::kLdsTopNodes, c->node_layout, c->sext, order, rank);
        if ((size_t)b->count + 1 > c->node_perm_cap) {
            uint32_t *old_order = c->node_order, *old_rank = c->node_rank;
            c->node_order = c->node_rank = nullptr, c->node_perm_cap = 0;
            if (old_order) HIP_TRY(hipFree(old_order));
            if (old_rank) HIP_TRY(hipFree(old_rank));
            HIP_TRY(hipMalloc((void **)&c->node_order, ((size_t)b->count + 1) * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void **)&c->node_rank, ((size_t)b->count + 1) * sizeof(uint32_t)));
            c->node_perm_cap = (size_t)b->count + 1;
        }
        if (b->count) {
            HIP_TRY(hipMemcpyAsync(c->node_order, order.data(), (size_t)b->count * 4, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->node_rank, rank.data(), (size_t)b->count * 4, hipMemcpyHostToDevice, c->stream));
            // the vectors die at the end of this block: the copies must have read them by then
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        rt::launch_repack_nodes(b->raw, b->count, c->smin, c->sext, c->node_order, c->node_rank, c->leaves, b->packed,
                                c->stream);
    }
    rt::launch_repack_tris(c->tris.raw, c->tris.count, c->tris.packed, c->stream);
    rt::launch_repack_spheres(c->spheres.raw, c->spheres.count, c->spheres.packed, c->stream);
    rt::launch_copy_raw32(c->boxes.raw, c->boxes.count, c->boxes.packed, c->stream);
    HIP_TRY(hipGetLastError());
    for (RawBuffer *b : part) b->dirty = false;
    return 0;
}

// builds the fast-mode tree if the primitives changed since the last build and copies it to the device
int prepare_fast(rt_ctx *c) {
    if (!c->fast_dirty) return 0;
    if (int rc = sync_all(c)) return rc;  // frames in flight still read the old tree
    c->main_dirty = true;
    // the primitives as uploaded, read back from the device (only a context that enables fast mode pays for this)
    std::vector<uint32_t> tris(c->tris.raw_bytes / 4), spheres(c->spheres.raw_bytes / 4), boxes(c->boxes.raw_bytes / 4);
    if (!tris.empty()) HIP_TRY(hipMemcpy(tris.data(), c->tris.raw, c->tris.raw_bytes, hipMemcpyDeviceToHost));
    if (!spheres.empty()) HIP_TRY(hipMemcpy(spheres.data(), c->spheres.raw, c->spheres.raw_bytes, hipMemcpyDeviceToHost));
    if (!boxes.empty()) HIP_TRY(hipMemcpy(boxes.data(), c->boxes.raw, c->boxes.raw_bytes, hipMemcpyDeviceToHost));
    if (rt::build_fast_bvh(tris.data(), (uint32_t)(tris.size() / 16), spheres.data(), (uint32_t)(spheres.size() / 8),
                                    boxes.data(), (uint32_t)(boxes.size() / 8), c->fbvh))
        return fail("rt_set_fast_mode: the scene's coordinates are too large for the fast-mode tree's quantised boxes");
    const size_t nb = c->fbvh.nodes.size() * 4;
    if (nb > c->fnodes_cap) {
        if (c->fnodes) HIP_TRY(hipFree(c->fnodes));
        c->fnodes = nullptr, c->fnodes_cap = 0;
        HIP_TRY(hipMalloc((void **)&c->fnodes, nb));
        c->fnodes_cap = nb;
    }
    HIP_TRY(hipMemcpy(c->fnodes, c->fbvh.nodes.data(), nb, hipMemcpyHostToDevice));
    c->fast_dirty = false;
    return 0;
}

int prepare_scene(rt_ctx *c, rt::DevScene *s) {
    if (int rc = repack_arena(c)) return rc;
    for (int binding : {RT_BINDING_PLANES, RT_BINDING_MATERIALS})
        if (int rc = repack(c, binding)) return rc;
    s->nodes = c->nodes.packed;
    s->tris = c->tris.packed;
    s->spheres = c->spheres.packed;
    s->boxes = c->boxes.packed;
    s->planes = c->planes.packed;
    s->mats = c->mats.packed;
    s->n_nodes = c->nodes.count;
    s->n_tris = c->tris.count;
    s->n_spheres = c->spheres.count;
    s->n_boxes = c->boxes.count;
    s->n_planes = c->planes.count;
    s->n_mats = c->mats.count;
    s->unit_tris = c->leaves.unit_tris, s->unit_spheres = c->leaves.unit_spheres, s->unit_boxes = c->leaves.unit_boxes;
    s->fnodes = nullptr;
    s->n_fnodes = 0;
    if (c->fast) {
        if (int rc = prepare_fast(c)) return rc;
        s->fnodes = c->fnodes;
        s->n_fnodes = c->fbvh.n_nodes;
    }
    memcpy(s->inv_vp, c->inv_vp, sizeof s->inv_vp);
    memcpy(s->eye, c->eye, sizeof s->eye);
    memcpy(s->smin, c->smin, sizeof s->smin);
    memcpy(s->sext, c->sext, sizeof s->sext);
    return 0;
}

void fill_params(rt_ctx *c, rt::TraceParams *p) {
    p->W = c->W;
    p->H = c->H;
    p->WW = (c->W / 8) * 8;  // glDispatchCompute(int(W/8), int(H/4), 1) with 8x4 groups
    p->HH = (c->H / 4) * 4;
    p->sample = c->sample;
    p->sum = c->sum;
    p->aux = nullptr;
    p->counters = c->counters;
    p->shard = 0;
    p->nshards = 1;
    p->tiles_x = (uint32_t)((p->WW + 15) / 16);
    p->tiles_y = (uint32_t)((p->HH + 15) / 16);
    p->ntiles = p->tiles_x * p->tiles_y;
    p->order = nullptr;
    p->cost = nullptr;
}

// number of macro tiles of the frame (the unit the megakernel deals to the XCDs)
uint32_t macro_tiles(const rt::TraceParams &p) {
    return ((p.tiles_x + RT_MACRO - 1u) / RT_MACRO) * ((p.tiles_y + RT_MACRO - 1u) / RT_MACRO);
}

// Queues the ordering pass for a frame of parity b on stream s and points the frame at its result.
int order_tiles(rt_ctx *c, rt::TraceParams *p, int b, bool serial, hipStream_t s) {
    p->order = nullptr, p->cost = nullptr;
    const uint32_t n = macro_tiles(*p);
    if (!c->lpt || c->pipeline != 0 || n > 8u * rt::kLptPerXcd) return 0;
    if (!serial)  // is a frame queued earlier still on the device?  then this one will not run alone
        for (int k = 0; k < 2; k++)
            if (c->busy_recorded[k]) {
                if (hipEventQuery(c->ev_busy[k]) == hipErrorNotReady) {
                    (void)hipGetLastError();  // "not ready" is an answer, not a failure to report later
                    return 0;
                }
                c->busy_recorded[k] = false;
            }
    // a frame on the main stream uses these buffers too: later frames on the trace streams must see it finished
    if (s == c->stream) c->main_dirty = true;
    rt::launch_lpt_sort(c->lpt_cost[b], c->lpt_order[b], n, s);
    p->order = c->lpt_order[b];
    p->cost = c->lpt_cost[b];
    return 0;
}

// pixels of the covered area that belong to `shard`: 64x64 macro tiles dealt round-robin in
// Morton order (the same rule as the kernels' tile_pixel)
uint32_t morton2(uint32_t x, uint32_t y) {
    auto part = [](uint32_t v) {
        v &= 0xFFFFu;
        v = (v | (v << 8)) & 0x00FF00FFu;
        v = (v | (v << 4)) & 0x0F0F0F0Fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return part(x) | (part(y) << 1);
}
uint64_t owned_pixels(int WW, int HH, uint32_t shard, uint32_t nshards) {
    if (nshards <= 1) return (uint64_t)WW * (uint64_t)HH;
    uint64_t n = 0;
    for (int ty = 0; ty * 64 < HH; ty++)
        for (int tx = 0; tx * 64 < WW; tx++)
            if (morton2((uint32_t)tx, (uint32_t)ty) % nshards == shard) {
                int w = WW - tx * 64 < 64 ? WW - tx * 64 : 64, h = HH - ty * 64 < 64 ? HH - ty * 64 : 64;
                n += (uint64_t)w * (uint64_t)h;
            }
    return n;
}

// Times the two traversal variants on this scene, image size and shard (see rt_ctx::leaf_steps) with
// launches that write no image and count their rays apart.  Called before the plain frames of a
// call (`frames` of them); runs once per scene, and not before the scene has rendered
// kTuneAfterFrames frames or a call of that many arrives: the four launches cost about four frames
// and the better variant wins 6 to 18 % of a frame, so a short job never earns them back.  Until then the size of the
// scene decides (cache-resident scenes do better with leaf steps of their own).
constexpr uint32_t kTuneAfterFrames = 64;
int choose_traversal_variant(rt_ctx *c, const rt::TraceParams &frame, uint32_t frames) {
    if (c->tuned || c->tune_fixed || c->pipeline != 0 || c->fast) return 0;
    const uint64_t records = (uint64_t)c->nodes.count * 2u + c->tris.count + c->spheres.count;  // in 64-byte units
    const uint64_t pixels = (uint64_t)c->W * (uint64_t)c->H;
    auto near = [](uint64_t a, uint64_t b) { return a <= b + b / 4u && b <= a + a / 4u; };
    if (c->tuned_records && near(records, c->tuned_records) && near(pixels, c->tuned_pixels)) {
        c->tuned = true;
        return 0;
    }
    if (c->plain_frames < kTuneAfterFrames && frames < kTuneAfterFrames) {
        c->leaf_steps = records < 100000u ? 1 : 0;
        c->plain_frames += frames;
        return 0;
    }
    c->tuned_records = records;
    c->tuned_pixels = pixels;
    if (int rc = join(c)) return rc;
    c->main_dirty = true;
    rt::TraceParams q = frame;
    q.sample = nullptr;
    q.sum = nullptr;
    q.aux = nullptr;
    q.counters = c->tune_counters;
    q.time = 0.0f;
    // a large unsharded image is timed on every k-th macro tile (the shard rule of rt_render_sum),
    // as long as the tiles left still fill the chip several times over
    if (q.nshards == 1u && q.ntiles >= 2u * 8160u) {
        q.nshards = q.ntiles / 8160u < 4u ? q.ntiles / 8160u : 4u;
        q.shard = 0u;
    }
    for (int k = 0; k < 4; k++) {
        HIP_TRY(hipEventRecord(c->ev_tune[k], c->stream));
        rt::launch_trace(q, 0, false, false, (k & 1) == 0, c->stream);
    }
    HIP_TRY(hipEventRecord(c->ev_tune[4], c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(c->ev_tune[4]));
    for (int k = 0; k < 4; k++) HIP_TRY(hipEventElapsedTime(&c->tune_ms[k], c->ev_tune[k], c->ev_tune[k + 1]));
    c->leaf_steps = c->tune_ms[2] < c->tune_ms[3] ? 1 : 0;
    c->tuned = true;
    return 0;
}

int begin_timed(rt_ctx *c) {
    HIP_TRY(hipEventRecord(c->ev_begin, c->stream));
    c->ev_trace_used = 0;
    c->tlast_used[0] = c->tlast_used[1] = false;
    return 0;
}

int trace_event(rt_ctx *c, hipStream_t s) {  // records the next event of the pool on stream s
    if (c->ev_trace_used >= c->ev_trace.size()) {
        if (c->ev_trace.size() >= 512) return 0;  // long runs: only the first 256 frames are split out
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->ev_trace.push_back(e);
    }
    HIP_TRY(hipEventRecord(c->ev_trace[c->ev_trace_used++], s));
    return 0;
}

int end_timed(rt_ctx *c) {
    HIP_TRY(hipEventRecord(c->ev_end, c->stream));
    c->timed = true;
    return 0;
}

}  // namespace

int rt::ctx_device(rt_ctx *c) { return c->device; }
void *rt::ctx_comm(rt_ctx *c) { return c->comm; }
void rt::ctx_set_comm(rt_ctx *c, void *comm, int rank, int world) {
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
}
double *rt::ctx_comm_scratch(rt_ctx *c) { return c->comm_scratch; }
int rt::ctx_join_for_collective(rt_ctx *c, hipStream_t *stream, float **own_sum, size_t *nfloats) {
    if (int rc = join(c)) return rc;
    c->main_dirty = true;  // the collective reads and writes images on `stream`
    *stream = c->stream;
    *own_sum = reinterpret_cast<float *>(c->sum);
    *nfloats = (size_t)c->W * c->H * 4;
    return 0;
}

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }

static int create_impl(rt_ctx *c, int device) {
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev_begin));
    HIP_TRY(hipEventCreate(&c->ev_end));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    for (int b = 0; b < 2; b++) {
        HIP_TRY(hipStreamCreateWithFlags(&c->tstream[b], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_tdone[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_acc[b], hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&c->ev_tlast[b]));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_busy[b], hipEventDisableTiming));
    }
    for (hipEvent_t &e : c->ev_tune) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipMalloc((void **)&c->tune_counters, sizeof(rt::Counters) * rt::kCounterStripes));
    HIP_TRY(hipMemsetAsync(c->tune_counters, 0, sizeof(rt::Counters) * rt::kCounterStripes, c->stream));
    if (const char *e = getenv("RT_OVERLAP")) c->overlap = atoi(e) != 0;
    if (const char *e = getenv("RT_LPT")) c->lpt = atoi(e) != 0;
    if (const char *e = getenv("RT_NODE_LAYOUT")) c->node_layout = atoi(e);
    for (int b = 0; b < 2; b++) {
        HIP_TRY(hipMalloc((void **)&c->lpt_cost[b], 8u * rt::kLptPerXcd * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&c->lpt_order[b], 8u * rt::kLptPerXcd * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(c->lpt_cost[b], 0, 8u * rt::kLptPerXcd * sizeof(uint32_t), c->stream));
    }
    if (const char *e = getenv("RT_LEAF_STEPS")) {
        c->leaf_steps = atoi(e) != 0;
        c->tune_fixed = true;
    }
    HIP_TRY(hipMalloc((void **)&c->counters, sizeof(rt::Counters) * rt::kCounterStripes));
    HIP_TRY(hipMemsetAsync(c->counters, 0, sizeof(rt::Counters) * rt::kCounterStripes, c->stream));
    HIP_TRY(hipMalloc((void **)&c->comm_scratch, sizeof(double) * RT_COMM_MAX_SCALARS));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    // persistent traversal kernel: one 64-lane workgroup per wave slot (32 per CU)
    c->extend_waves = (uint32_t)prop.multiProcessorCount * 32u;
    // experiment knobs (defaults are the tuned values)
    if (const char *e = getenv("RT_PIPELINE")) c->pipeline = (strcmp(e, "wave") == 0) ? 1 : (strcmp(e, "regroup") == 0) ? 2 : 0;
    if (const char *e = getenv("RT_REFILL")) c->refill_threshold = (uint32_t)atoi(e);
    if (const char *e = getenv("RT_WAVE_BIN")) c->wave_bin = atoi(e) != 0;
    if (const char *e = getenv("RT_EXTEND_WAVES")) c->extend_waves = (uint32_t)atoi(e);
    if (c->refill_threshold < 1u) c->refill_threshold = 1u;
    if (c->refill_threshold > 64u) c->refill_threshold = 64u;
    if (c->extend_waves < 1u) c->extend_waves = 1u;
    HIP_TRY(rt::noalloc_prepare_device(device));
    return alloc_images(c, c->W, c->H);
}

int rt_create(int device, rt_ctx **out) {
    if (!out) return fail("rt_create: out is null");
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail("rt_create: no such HIP device: " + std::to_string(device) + " (" + std::to_string(ndev) +
                    " visible)");
    HIP_TRY(hipSetDevice(device));
    rt_ctx *c = new rt_ctx();
    if (int rc = create_impl(c, device)) {
        const std::string why = g_err;  // rt_destroy releases whatever was created so far
        rt_destroy(c);
        g_err = why;
        return rc;
    }
    *out = c;
    return 0;
}

// Tolerates a partially constructed context (null streams, events and buffers).
int rt_destroy(rt_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    for (int b = 0; b < 2; b++)
        if (c->tstream[b]) (void)hipStreamSynchronize(c->tstream[b]);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    rt::comm_destroy(c->comm);
    if (c->comm_scratch) (void)hipFree(c->comm_scratch);
    for (RawBuffer *b : {&c->spheres, &c->planes, &c->boxes, &c->mats, &c->nodes, &c->tris})
        if (b->raw) (void)hipFree(b->raw);
    for (RawBuffer *b : {&c->planes, &c->mats})
        if (b->packed) (void)hipFree(b->packed);
    if (c->arena) (void)hipFree(c->arena);
    if (c->fnodes) (void)hipFree(c->fnodes);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (void *p : {(void *)c->lpt_cost[0], (void *)c->lpt_cost[1], (void *)c->lpt_order[0], (void *)c->lpt_order[1]})
        if (p) (void)hipFree(p);
    for (void *p : {(void *)c->node_order, (void *)c->node_rank, (void *)c->sample, (void *)c->sample2, (void *)c->acc,
                    (void *)c->sum, (void *)c->sum2, (void *)c->aux, (void *)c->counters, (void *)c->q.rays[0],
                    (void *)c->q.hits, (void *)c->q.paths})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : c->ev_trace) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_tune)
        if (e) (void)hipEventDestroy(e);
    if (c->tune_counters) (void)hipFree(c->tune_counters);
    for (hipEvent_t e : {c->ev_begin, c->ev_end, c->ev_ready, c->ev_tdone[0], c->ev_tdone[1], c->ev_acc[0], c->ev_acc[1],
                         c->ev_tlast[0], c->ev_tlast[1], c->ev_busy[0], c->ev_busy[1]})
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t st : {c->tstream[0], c->tstream[1], c->stream})
        if (st) (void)hipStreamDestroy(st);
    delete c;
    return 0;
}

int rt_resize(rt_ctx *c, int width, int height) {
    if (!c) return fail("rt_resize: null context");
    if (width <= 0 || height <= 0) return fail("rt_resize: size must be positive");
    if ((size_t)width * (size_t)height > ((size_t)1 << 28)) return fail("rt_resize: image too large");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = sync_all(c)) return rc;
    c->tuned = false, c->plain_frames = 0;
    return alloc_images(c, width, height);
}

int rt_upload(rt_ctx *c, int binding, const void *data, size_t nbytes) {
    if (!c) return fail("rt_upload: null context");
    RawBuffer *b = buffer_of(c, binding);
    if (!b) return fail("rt_upload: unknown binding " + std::to_string(binding));
    if (nbytes && !data) return fail("rt_upload: null data");
    int stride = stride_of(binding);
    if (nbytes % (size_t)stride) return fail("rt_upload: size is not a multiple of the element stride");
    // 30-bit indices; for nodes the index one past the array (the sentinel) must be below 0x3FFFFFFF too
    if (nbytes / (size_t)stride >= ((size_t)1 << 30) - (binding == RT_BINDING_BVH ? 1u : 0u))
        return fail("rt_upload: too many elements");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = sync_all(c)) return rc;  // the old buffer may still be in use
    // From here until the copy has succeeded the binding is empty (and its host mirrors with it), so
    // that a failed upload leaves a scene without these elements rather than counts that no longer
    // match the device records.
    uint32_t *old_raw = b->raw;
    b->raw = nullptr;
    b->raw_bytes = 0;
    b->count = 0;
    b->dirty = true;
    if (binding != RT_BINDING_MATERIALS) c->tuned = false, c->plain_frames = 0;  // a new scene: time the variants again
    // host copy of the node words (the breadth-first renumbering walks them); the fast-mode tree reads
    // the primitives back from the device when it is first needed (prepare_fast)
    std::vector<uint32_t> *mirror = binding == RT_BINDING_BVH ? &c->host_nodes : nullptr;
    if (mirror) mirror->clear();
    if (binding == RT_BINDING_TRIANGLES || binding == RT_BINDING_SPHERES || binding == RT_BINDING_BOXES)
        c->fast_dirty = true;
    if (old_raw) HIP_TRY(hipFree(old_raw));
    if (nbytes) {
        uint32_t *raw = nullptr;
        HIP_TRY(hipMalloc((void **)&raw, nbytes));
        // the caller's array may be reused as soon as this returns, as with glBufferData
        hipError_t e = hipMemcpy(raw, data, nbytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(raw);
            return fail(std::string("rt_upload: copy to the device: ") + hipGetErrorString(e));
        }
        if (mirror) mirror->assign((const uint32_t *)data, (const uint32_t *)data + nbytes / 4);
        b->raw = raw;
        b->raw_bytes = nbytes;
        b->count = (uint32_t)(nbytes / (size_t)stride);
    }
    return 0;
}

int rt_set_scene_bounds(rt_ctx *c, const float mn[3], const float ext[3]) {
    if (!c || !mn || !ext) return fail("rt_set_scene_bounds: null argument");
    if (memcmp(c->smin, mn, 12) || memcmp(c->sext, ext, 12)) c->nodes.dirty = true;  // boxes are decoded with them
    memcpy(c->smin, mn, 12);
    memcpy(c->sext, ext, 12);
    return 0;
}

int rt_set_camera(rt_ctx *c, const float m[16], const float eye[3]) {
    if (!c || !m || !eye) return fail("rt_set_camera: null argument");
    memcpy(c->inv_vp, m, 64);
    memcpy(c->eye, eye, 12);
    return 0;
}

int rt_render(rt_ctx *c, uint32_t first_frame, uint32_t nframes, float bounces, uint32_t flags) {
    if (!c) return fail("rt_render: null context");
    if (!(bounces >= 0.0f)) return fail("rt_render: bounces must not be negative");  // NaN included
    HIP_TRY(hipSetDevice(c->device));
    rt::TraceParams p;
    if (int rc = prepare_scene(c, &p.scene)) return rc;
    fill_params(c, &p);
    size_t npix = (size_t)c->W * c->H;
    const bool aux = flags & RT_FLAG_AUX, stats = flags & RT_FLAG_STATS;
    // frames that write the per-pixel record, and the wavefront pipeline (one ray queue), run one at a time
    const bool serial = !c->overlap || aux || c->pipeline == 1;
    if (c->tpending[0] || c->tpending[1] || c->sum2_dirty)
        if (int rc = join(c)) return rc;  // sum-mode frames still in flight
    if (aux && !c->aux) {
        HIP_TRY(hipMalloc((void **)&c->aux, npix * 16 * sizeof(uint32_t)));
    }
    if (aux) HIP_TRY(hipMemsetAsync(c->aux, 0, npix * 16 * sizeof(uint32_t), c->stream));
    p.aux = aux ? c->aux : nullptr;
    p.bounces = bounces;
    if (!aux && !stats && nframes)
        if (int rc = choose_traversal_variant(c, p, nframes)) return rc;
    if (int rc = begin_timed(c)) return rc;
    for (uint32_t i = 0; i < nframes; i++) {
        p.time = (float)(first_frame + i);  // renderer.py:211 float(self._frames)
        const int b = c->flip;
        c->flip ^= 1;
        p.sample = b ? c->sample2 : c->sample;
        hipStream_t ts;
        if (int rc = trace_stream(c, b, serial, &ts)) return rc;
        // the accumulate pass that last read this sample image must be done before it is rewritten
        if (ts != c->stream && c->acc_recorded[b]) HIP_TRY(hipStreamWaitEvent(ts, c->ev_acc[b], 0));
        if (int rc = order_tiles(c, &p, b, serial, ts)) return rc;
        if (int rc = trace_event(c, ts)) return rc;
        trace_frame(c, p, 0, aux, stats, ts);
        if (ts != c->stream) {
            HIP_TRY(hipEventRecord(c->ev_busy[b], ts));
            c->busy_recorded[b] = true;
        }
        c->pixel_samples += (uint64_t)p.WW * (uint64_t)p.HH;
        if (int rc = trace_event(c, ts)) return rc;
        if (ts != c->stream) {
            HIP_TRY(hipEventRecord(c->ev_tdone[b], ts));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tdone[b], 0));
        }
        rt::launch_accumulate(p.sample, c->acc, npix, (float)c->rendered_frames, c->stream);
        HIP_TRY(hipEventRecord(c->ev_acc[b], c->stream));
        c->acc_recorded[b] = true;
        c->last_sample = b;
        c->rendered_frames++;
    }
    HIP_TRY(hipGetLastError());
    return end_timed(c);
}

int rt_reset_accumulation(rt_ctx *c) {
    if (!c) return fail("rt_reset_accumulation: null context");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = join(c)) return rc;
    HIP_TRY(hipMemcpyAsync(c->acc, c->last_sample ? c->sample2 : c->sample, (size_t)c->W * c->H * sizeof(float4),
                           hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->ev_acc[c->last_sample], c->stream));
    c->acc_recorded[c->last_sample] = true;
    c->rendered_frames = 0;
    return 0;
}

int rt_set_rendered_frames(rt_ctx *c, uint32_t frames) {
    if (!c) return fail("rt_set_rendered_frames: null context");
    c->rendered_frames = frames;  // read on the host when a frame is queued: frames already queued keep theirs
    return 0;
}

int rt_rendered_frames(rt_ctx *c, uint32_t *out) {
    if (!c || !out) return fail("rt_rendered_frames: null argument");
    *out = c->rendered_frames;
    return 0;
}

static int read_image(rt_ctx *c, const void *dev, void *host, size_t bytes, const char *what) {
    if (!c || !host) return fail(std::string(what) + ": null argument");
    if (!dev) return fail(std::string(what) + ": nothing has been rendered into that buffer");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = join(c)) return rc;
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rt_read_sample(rt_ctx *c, float *rgba) {
    return read_image(c, c ? (c->last_sample ? c->sample2 : c->sample) : nullptr, rgba, c ? (size_t)c->W * c->H * 16 : 0, "rt_read_sample");
}
int rt_read_accumulation(rt_ctx *c, float *rgba) {
    return read_image(c, c ? c->acc : nullptr, rgba, c ? (size_t)c->W * c->H * 16 : 0, "rt_read_accumulation");
}
int rt_read_aux(rt_ctx *c, uint32_t *aux16) {
    return read_image(c, c ? c->aux : nullptr, aux16, c ? (size_t)c->W * c->H * 64 : 0, "rt_read_aux");
}

int rt_map_image(rt_ctx *c, int which, const float **rgba) {
    if (!c || !rgba) return fail("rt_map_image: null argument");
    if (which != RT_IMAGE_SAMPLE && which != RT_IMAGE_ACCUMULATION) return fail("rt_map_image: unknown image");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->W * c->H * sizeof(float4);
    if (!c->pinned) {
        HIP_TRY(hipHostMalloc((void **)&c->pinned, bytes, hipHostMallocDefault));
        c->pinned_bytes = bytes;
    }
    const float4 *dev = which == RT_IMAGE_ACCUMULATION ? c->acc : (c->last_sample ? c->sample2 : c->sample);
    if (int rc = join(c)) return rc;
    HIP_TRY(hipMemcpyAsync(c->pinned, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *rgba = c->pinned;
    return 0;
}

int rt_sync(rt_ctx *c) {
    if (!c) return fail("rt_sync: null context");
    HIP_TRY(hipSetDevice(c->device));
    return sync_all(c);
}

int rt_get_stats(rt_ctx *c, rt_stats *out) {
    if (!c || !out) return fail("rt_get_stats: null argument");
    HIP_TRY(hipSetDevice(c->device));
    rt::Counters hs[rt::kCounterStripes];
    if (int rc = join(c)) return rc;
    HIP_TRY(hipMemcpyAsync(hs, c->counters, sizeof hs, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memset(out, 0, sizeof *out);
    for (const rt::Counters &h : hs) {
        out->rays += h.rays;
        out->nodes += h.nodes;
        out->tri_tests += h.tri_tests;
        out->sphere_tests += h.sphere_tests;
        out->box_tests += h.box_tests;
        if (h.max_stack > out->max_stack) out->max_stack = h.max_stack;
        out->stack_overflows += h.stack_overflows;
        out->hits += h.hits;
    }
    out->pixel_samples = c->pixel_samples;  // known on the host: every launch covers a fixed pixel set
    out->last_render_ms = 0.0;
    out->last_trace_ms = 0.0;
    if (c->timed) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_begin, c->ev_end));
        out->last_render_ms = ms;
        // overlapped sum-mode launches run on the trace streams, which `stream` (where ev_end was
        // recorded) had not joined yet: the call ends when the last of them does
        for (int b = 0; b < 2; b++)
            if (c->tlast_used[b]) {
                HIP_TRY(hipEventElapsedTime(&ms, c->ev_begin, c->ev_tlast[b]));
                if (ms > out->last_render_ms) out->last_render_ms = ms;
            }
        double t = 0.0;
        for (size_t i = 0; i + 1 < c->ev_trace_used; i += 2) {
            HIP_TRY(hipEventElapsedTime(&ms, c->ev_trace[i], c->ev_trace[i + 1]));
            t += ms;
        }
        out->last_trace_ms = t;
    }
    return 0;
}

int rt_reset_stats(rt_ctx *c) {
    if (!c) return fail("rt_reset_stats: null context");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = join(c)) return rc;
    c->main_dirty = true;
    HIP_TRY(hipMemsetAsync(c->counters, 0, sizeof(rt::Counters) * rt::kCounterStripes, c->stream));
    c->pixel_samples = 0;
    return 0;
}

int rt_render_sum(rt_ctx *c, uint32_t first_frame, uint32_t nframes, float bounces, uint32_t shard,
                  uint32_t nshards, uint32_t frame_stride, uint32_t frame_phase, float *sum_rgba) {
    if (!c) return fail("rt_render_sum: null context");
    if (nshards == 0 || shard >= nshards) return fail("rt_render_sum: shard must be < nshards");
    if (frame_stride == 0 || frame_phase >= frame_stride) return fail("rt_render_sum: frame_phase must be < frame_stride");
    if (!(bounces >= 0.0f)) return fail("rt_render_sum: bounces must not be negative");
    HIP_TRY(hipSetDevice(c->device));
    rt::TraceParams p;
    if (int rc = prepare_scene(c, &p.scene)) return rc;
    fill_params(c, &p);
    p.sample = nullptr;
    float4 *target = sum_rgba ? reinterpret_cast<float4 *>(sum_rgba) : c->sum;
    p.shard = shard;
    p.nshards = nshards;
    p.bounces = bounces;
    const uint64_t owned = owned_pixels(p.WW, p.HH, shard, nshards);
    const bool serial = !c->overlap || c->pipeline == 1;
    // odd frames are summed in a second image and folded into the target by the next join; a
    // different target starts a new pair
    if (c->sum2_dirty && c->sum2_target != target)
        if (int rc = join(c)) return rc;
    if (frame_phase < nframes)
        if (int rc = choose_traversal_variant(c, p, (nframes - frame_phase + frame_stride - 1u) / frame_stride)) return rc;
    if (int rc = begin_timed(c)) return rc;
    for (uint32_t i = frame_phase; i < nframes; i += frame_stride) {
        p.time = (float)(first_frame + i);
        const int b = serial ? 0 : c->flip;
        if (!serial) c->flip ^= 1;
        hipStream_t ts;
        if (int rc = trace_stream(c, b, serial, &ts)) return rc;
        p.sum = (b && !serial) ? c->sum2 : target;
        if (int rc = order_tiles(c, &p, b, serial, ts)) return rc;
        if (int rc = trace_event(c, ts)) return rc;
        trace_frame(c, p, 1, false, false, ts);
        if (ts != c->stream) {
            HIP_TRY(hipEventRecord(c->ev_busy[b], ts));
            c->busy_recorded[b] = true;
        }
        c->pixel_samples += owned;
        if (int rc = trace_event(c, ts)) return rc;
        if (ts != c->stream) {
            HIP_TRY(hipEventRecord(c->ev_tlast[b], ts));
            c->tlast_used[b] = true;
            c->tpending[b] = true;
            if (b) {
                c->sum2_dirty = true;
                c->sum2_target = target;
            }
        }
    }
    HIP_TRY(hipGetLastError());
    return end_timed(c);
}

int rt_set_fast_mode(rt_ctx *c, int enable) {
    if (!c) return fail("rt_set_fast_mode: null context");
    HIP_TRY(hipSetDevice(c->device));
    if ((enable != 0) != c->fast)
        if (int rc = sync_all(c)) return rc;
    c->fast = enable != 0;
    return 0;
}

int rt_fast_bvh_size(rt_ctx *c, uint32_t *n_nodes) {
    if (!c || !n_nodes) return fail("rt_fast_bvh_size: null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = prepare_fast(c)) return rc;
    *n_nodes = c->fbvh.n_nodes;
    return 0;
}

int rt_fast_bvh_export(rt_ctx *c, void *nodes, size_t nodes_bytes) {
    if (!c || !nodes) return fail("rt_fast_bvh_export: null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = prepare_fast(c)) return rc;
    if (nodes_bytes != c->fbvh.nodes.size() * 4)
        return fail("rt_fast_bvh_export: the buffer must hold 64 bytes times (the count rt_fast_bvh_size reports + 1)");
    memcpy(nodes, c->fbvh.nodes.data(), nodes_bytes);
    return 0;
}

int rt_traversal_variant(rt_ctx *c, int *leaf_steps, int *decided) {
    if (!c || !leaf_steps || !decided) return fail("rt_traversal_variant: null argument");
    *leaf_steps = c->leaf_steps;
    *decided = (c->tune_fixed || c->tuned) ? 1 : 0;
    return 0;
}

int rt_set_traversal_variant(rt_ctx *c, int variant) {
    if (!c) return fail("rt_set_traversal_variant: null context");
    if (variant < -1 || variant > 1) return fail("rt_set_traversal_variant: variant must be -1, 0 or 1");
    if (variant < 0) {
        c->tune_fixed = false;
        c->tuned = false, c->tuned_records = 0, c->plain_frames = 0;
    } else {
        c->leaf_steps = variant;
        c->tune_fixed = true;
    }
    return 0;
}

int rt_device_info(rt_ctx *c, int *device, char *pci_bus_id, size_t pci_bus_id_bytes) {
    if (!c || !device || !pci_bus_id) return fail("rt_device_info: null argument");
    if (pci_bus_id_bytes < 16) return fail("rt_device_info: pci_bus_id must hold at least 16 bytes");
    *device = c->device;
    HIP_TRY(hipDeviceGetPCIBusId(pci_bus_id, (int)pci_bus_id_bytes, c->device));
    return 0;
}

int rt_shader_clock(rt_ctx *c, double *ghz) {
    if (!c || !ghz) return fail("rt_shader_clock: null argument");
    HIP_TRY(hipSetDevice(c->device));
    rt::Counters hs[rt::kCounterStripes];
    if (int rc = join(c)) return rc;
    HIP_TRY(hipMemcpyAsync(hs, c->counters, sizeof hs, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long cycles = 0, ticks = 0;
    for (const rt::Counters &h : hs) cycles += h.clk_cycles, ticks += h.clk_ticks;
    *ghz = ticks ? (double)cycles / (double)ticks * 0.1 : 0.0;  // ticks of the constant 100 MHz counter
    return 0;
}

int rt_selftest(rt_ctx *c, unsigned long long *reciprocal_mismatches) {
    if (!c || !reciprocal_mismatches) return fail("rt_selftest: null argument");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = join(c)) return rc;
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof *d));
    hipError_t e = hipMemsetAsync(d, 0, sizeof *d, c->stream);
    if (e == hipSuccess) {
        rt::launch_reciprocal_selftest(d, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(reciprocal_mismatches, d, sizeof *d, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(std::string("rt_selftest: ") + hipGetErrorString(e));
    return 0;
}

int rt_sum_buffer(rt_ctx *c, float **device_ptr, size_t *nbytes) {
    if (!c || !device_ptr || !nbytes) return fail("rt_sum_buffer: null argument");
    *device_ptr = reinterpret_cast<float *>(c->sum);
    *nbytes = (size_t)c->W * c->H * sizeof(float4);
    return 0;
}

int rt_clear_sum(rt_ctx *c, float *sum_rgba) {
    if (!c) return fail("rt_clear_sum: null context");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = join(c)) return rc;
    c->main_dirty = true;
    c->flip = 0;  // the launch after a clear adds into the sum image itself, the one after it into the second image
    HIP_TRY(hipMemsetAsync(sum_rgba ? (void *)sum_rgba : (void *)c->sum, 0, (size_t)c->W * c->H * sizeof(float4),
                           c->stream));
    return 0;
}

int rt_finalize_sum(rt_ctx *c, const float *sum_rgba, uint32_t spp) {
    if (!c) return fail("rt_finalize_sum: null context");
    if (spp == 0) return fail("rt_finalize_sum: spp must be positive");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = join(c)) return rc;
    rt::launch_finalize_sum(sum_rgba ? reinterpret_cast<const float4 *>(sum_rgba) : c->sum, c->acc,
                            (size_t)c->W * c->H, spp, c->stream);
    HIP_TRY(hipGetLastError());
    c->main_dirty = true;  // a later overlapped frame adds into the image this pass reads
    return 0;
}

}  // extern "C"
