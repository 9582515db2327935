// librt_scene.so — host-side BVH construction for the render path's node buffer (binding 11).
//
// Produces, byte for byte, the buffer the reference's Python builder produces for the same
// primitive list: binned-SAH binary tree (reference src/bvh/bvh_node.py:103-230), collapse of two
// binary levels into 4-wide nodes (:374-405), depth-first pre-order flattening with 16-bit
// quantised child boxes (:407-503) and the 64-byte node packing (:505-556).  The reference runs
// this in CPython with numpy float32 scalars; every arithmetic step below is therefore a
// separately rounded binary32 operation in the same order (build with -ffp-contract=off).
//
// Pure host code (no GPU): it feeds rt_upload(RT_BINDING_BVH, ...).

#include "rt_fastbvh.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

struct Prim {
    float mn[3], mx[3];
    uint32_t index;
    uint32_t type;  // 0 triangle, 1 sphere, 2 box
};

struct BNode {
    float mn[3], mx[3];
    int left = -1, right = -1;  // binary children, -1 for a leaf
    uint32_t prim_index = 0, prim_type = 3;
};

struct Box3 {
    float mn[3], mx[3];
    bool valid = false;
    void grow(const float *a, const float *b) {
        if (!valid) {
            for (int k = 0; k < 3; k++) { mn[k] = a[k]; mx[k] = b[k]; }
            valid = true;
        } else {
            for (int k = 0; k < 3; k++) { mn[k] = std::fmin(mn[k], a[k]); mx[k] = std::fmax(mx[k], b[k]); }
        }
    }
    // AABB.surface_area (reference src/aabb/aabb.py:61-70) on float32 values
    float area() const {
        float d0 = mx[0] - mn[0], d1 = mx[1] - mn[1], d2 = mx[2] - mn[2];
        return 2.0f * ((d0 * d1 + d0 * d2) + d1 * d2);
    }
};

struct Builder {
    const std::vector<Prim> &prims;
    std::vector<int> order;  // working permutation of primitive ids
    std::vector<BNode> nodes;
    static constexpr int kBuckets = 12;

    explicit Builder(const std::vector<Prim> &p) : prims(p) {
        order.resize(p.size());
        for (size_t i = 0; i < p.size(); i++) order[i] = (int)i;
        nodes.reserve(p.size() * 2 + 1);
    }

    // bvh_node.py:103-230 on order[lo, hi)
    int build(int lo, int hi, int depth) {
        int n = hi - lo;
        if (n <= 1) {
            const Prim &p = prims[order[lo]];
            BNode b;
            memcpy(b.mn, p.mn, 12);
            memcpy(b.mx, p.mx, 12);
            b.prim_index = p.index;
            b.prim_type = p.type;
            nodes.push_back(b);
            return (int)nodes.size() - 1;
        }
        const int axis = depth % 3;
        Box3 global;
        for (int i = lo; i < hi; i++) global.grow(prims[order[i]].mn, prims[order[i]].mx);
        const float axis_min = global.mn[axis], axis_max = global.mx[axis];
        const float axis_range = (axis_max > axis_min) ? (axis_max - axis_min) : 1e-8f;

        int count[kBuckets] = {0};
        Box3 bounds[kBuckets];
        for (int i = lo; i < hi; i++) {
            const Prim &p = prims[order[i]];
            float centroid = (p.mn[axis] + p.mx[axis]) * 0.5f;
            float f = ((centroid - axis_min) / axis_range) * (float)(kBuckets - 1);
            int b = (int)f;  // int() truncates toward zero
            if (b < 0) b = 0;  // a Python negative index would wrap; unreachable for finite boxes
            if (b > kBuckets - 1) b = kBuckets - 1;
            count[b]++;
            bounds[b].grow(p.mn, p.mx);
        }
        Box3 prefix_b[kBuckets], suffix_b[kBuckets];
        int prefix_c[kBuckets], suffix_c[kBuckets];
        for (int i = 0; i < kBuckets; i++) {
            if (count[i] == 0) {
                if (i > 0) { prefix_b[i] = prefix_b[i - 1]; prefix_c[i] = prefix_c[i - 1]; }
                else prefix_c[i] = 0;
            } else {
                if (i > 0 && prefix_b[i - 1].valid) prefix_b[i] = prefix_b[i - 1];
                prefix_b[i].grow(bounds[i].mn, bounds[i].mx);
                prefix_c[i] = count[i] + (i > 0 ? prefix_c[i - 1] : 0);
            }
        }
        for (int i = kBuckets - 1; i >= 0; i--) {
            if (count[i] == 0) {
                if (i < kBuckets - 1) { suffix_b[i] = suffix_b[i + 1]; suffix_c[i] = suffix_c[i + 1]; }
                else suffix_c[i] = 0;
            } else {
                if (i < kBuckets - 1 && suffix_b[i + 1].valid) suffix_b[i] = suffix_b[i + 1];
                suffix_b[i].grow(bounds[i].mn, bounds[i].mx);
                suffix_c[i] = count[i] + (i < kBuckets - 1 ? suffix_c[i + 1] : 0);
            }
        }
        float best_cost = std::numeric_limits<float>::infinity();
        int best_split = -1;
        const float sa_p = global.area();
        for (int i = 1; i < kBuckets; i++) {
            int lc = prefix_c[i - 1], rc = suffix_c[i];
            if (lc == 0 || rc == 0) continue;
            float cost = (prefix_b[i - 1].area() / sa_p) * (float)lc + (suffix_b[i].area() / sa_p) * (float)rc;
            if (cost < best_cost) {
                best_cost = cost;
                best_split = i;
            }
        }
        int mid;
        if (best_split < 0) {
            // fallback: stable sort by the box minimum on the axis, split in the middle
            std::stable_sort(order.begin() + lo, order.begin() + hi,
                             [&](int a, int b) { return prims[a].mn[axis] < prims[b].mn[axis]; });
            mid = n / 2;
        } else {
            std::stable_sort(order.begin() + lo, order.begin() + hi, [&](int a, int b) {
                float ka = 0.5f * (prims[a].mn[axis] + prims[a].mx[axis]);
                float kb = 0.5f * (prims[b].mn[axis] + prims[b].mx[axis]);
                return ka < kb;
            });
            mid = 0;
            for (int i = 0; i < best_split; i++) mid += count[i];
        }
        int l = build(lo, lo + mid, depth + 1);
        int r = build(lo + mid, hi, depth + 1);
        BNode b;
        for (int k = 0; k < 3; k++) {
            b.mn[k] = std::fmin(nodes[l].mn[k], nodes[r].mn[k]);
            b.mx[k] = std::fmax(nodes[l].mx[k], nodes[r].mx[k]);
        }
        b.left = l;
        b.right = r;
        nodes.push_back(b);
        return (int)nodes.size() - 1;
    }
};

// float -> uint16 the way numpy's astype(np.uint16) behaves on x86-64 (truncating convert to a
// 32-bit integer, low 16 bits kept; NaN and out-of-range give 0)
inline uint32_t to_u16(float v) {
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return 0u;
    return (uint32_t)((int32_t)v) & 0xFFFFu;
}

struct Flattener {
    const std::vector<BNode> &b;
    std::vector<uint32_t> out;  // 16 words per node
    float smin[3], sext[3];

    bool is_leaf(int i) const { return b[i].left < 0; }

    // QBVHNode.from_bvh candidates (bvh_node.py:385-405) + flatten_qbvh.assign_index (:415-496)
    uint32_t emit(int node) {
        uint32_t idx = (uint32_t)(out.size() / 16);
        out.resize(out.size() + 16, 0u);
        int cand[4] = {-1, -1, -1, -1};
        int nc = 0;
        if (!is_leaf(node)) {
            for (int child : {b[node].left, b[node].right}) {
                if (is_leaf(child)) {
                    if (nc < 4) cand[nc++] = child;
                } else {
                    if (nc < 4) cand[nc++] = b[child].left;
                    if (nc < 4) cand[nc++] = b[child].right;
                }
            }
        }
        uint32_t words[16] = {0};
        for (int s = 0; s < 4; s++) {
            int c = cand[s];
            if (c < 0) {
                words[12 + s] = 0xFFFFFFFFu;  // empty slot: boxes 0, type 3, index all ones
                continue;
            }
            uint32_t q[6];
            for (int k = 0; k < 3; k++) {
                float nmin = (b[c].mn[k] - smin[k]) / sext[k];
                float nmax = (b[c].mx[k] - smin[k]) / sext[k];
                q[k] = to_u16(std::nearbyintf(nmin * 65535.0f));
                q[3 + k] = to_u16(std::nearbyintf(nmax * 65535.0f));
            }
            for (int k = 0; k < 3; k++) words[4 * k + s] = (q[3 + k] << 16) | q[k];
            if (is_leaf(c)) {
                words[12 + s] = (b[c].prim_index & 0x3FFFFFFFu) | ((b[c].prim_type & 3u) << 30);
            } else {
                uint32_t ci = emit(c);  // children are numbered in slot order, depth first
                words[12 + s] = (ci & 0x3FFFFFFFu) | (3u << 30);
            }
        }
        memcpy(out.data() + (size_t)idx * 16, words, sizeof words);
        return idx;
    }
};

}  // namespace

extern "C" {

const char *rts_last_error(void) { return g_err.c_str(); }

// Builds the flattened 4-wide BVH for `n` primitives given in the order the reference's loader
// appends them (spheres, then quad triangles, then mesh triangles; src/scene/scene.py:27-30).
//   aabb_min / aabb_max : n*3 float32 primitive boxes (Primitive.aabb)
//   types               : n entries, 0 triangle / 1 sphere / 2 box
//   indices             : n entries, index into the primitive's own buffer
// On success *nodes_out is a malloc'ed array of *n_nodes_out * 16 uint32 (free with rts_free),
// and scene_min / scene_extent receive the root box (scene.py:70-71).
int rts_build_qbvh(const float *aabb_min, const float *aabb_max, const uint8_t *types, const uint32_t *indices,
                   uint32_t n, uint32_t **nodes_out, uint32_t *n_nodes_out, float scene_min[3],
                   float scene_extent[3]) {
    if (!nodes_out || !n_nodes_out || !scene_min || !scene_extent) {
        g_err = "rts_build_qbvh: null output argument";
        return 1;
    }
    if (n == 0) {
        g_err = "rts_build_qbvh: no primitives (the reference's builder fails on an empty scene too)";
        return 1;
    }
    if (!aabb_min || !aabb_max || !types || !indices) {
        g_err = "rts_build_qbvh: null input argument";
        return 1;
    }
    std::vector<Prim> prims(n);
    for (uint32_t i = 0; i < n; i++) {
        memcpy(prims[i].mn, aabb_min + (size_t)i * 3, 12);
        memcpy(prims[i].mx, aabb_max + (size_t)i * 3, 12);
        prims[i].index = indices[i];
        prims[i].type = types[i];
    }
    Builder bld(prims);
    int root = bld.build(0, (int)n, 0);
    Flattener fl{bld.nodes, {}, {0, 0, 0}, {0, 0, 0}};
    for (int k = 0; k < 3; k++) {
        fl.smin[k] = bld.nodes[root].mn[k];
        fl.sext[k] = bld.nodes[root].mx[k] - bld.nodes[root].mn[k];
        scene_min[k] = fl.smin[k];
        scene_extent[k] = fl.sext[k];
    }
    fl.out.reserve((size_t)n * 16);
    fl.emit(root);
    size_t bytes = fl.out.size() * sizeof(uint32_t);
    uint32_t *buf = (uint32_t *)malloc(bytes);
    if (!buf) {
        g_err = "rts_build_qbvh: out of memory";
        return 1;
    }
    memcpy(buf, fl.out.data(), bytes);
    *nodes_out = buf;
    *n_nodes_out = (uint32_t)(fl.out.size() / 16);
    return 0;
}

void rts_free(void *p) { free(p); }

// The fast-mode tree (rt_fastbvh.cpp) for reference-layout triangle / sphere / box buffers, without
// a GPU: what rt_set_fast_mode makes the render library build internally.  Output: a malloc'ed
// array of 16 words per node, *n_nodes_out nodes followed by one sentinel (free with rts_free).
int rts_build_fast_bvh(const uint32_t *tris, uint32_t n_tris, const uint32_t *spheres, uint32_t n_spheres,
                       const uint32_t *boxes, uint32_t n_boxes, uint32_t **nodes_out, uint32_t *n_nodes_out) {
    if (!nodes_out || !n_nodes_out) {
        g_err = "rts_build_fast_bvh: null output";
        return 1;
    }
    rt::FastBvh b;
    if (rt::build_fast_bvh(tris, n_tris, spheres, n_spheres, boxes, n_boxes, b)) {
        g_err = "rts_build_fast_bvh: the scene's coordinates are too large for conservative 8-bit child boxes";
        return 1;
    }
    uint32_t *n = (uint32_t *)malloc(b.nodes.size() * 4);
    if (!n) {
        g_err = "rts_build_fast_bvh: out of memory";
        return 1;
    }
    memcpy(n, b.nodes.data(), b.nodes.size() * 4);
    *nodes_out = n, *n_nodes_out = b.n_nodes;
    return 0;
}

}  // extern "C"
