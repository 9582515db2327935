// Builder of the acceleration structure of the opt-in "fast mode" (host side, plain C++).
//
// Fast mode is NOT the reference's data: the reference's tree (src/bvh/bvh_node.py:103-230: binned
// SAH on the axis depth % 3; :476-480: child boxes rounded to the NEAREST 16-bit code, i.e. not
// conservative) decides which hits its traversal finds, and the default mode consumes exactly
// that tree.  Here the library builds its own tree over the same uploaded primitives and the
// traversal (same visiting rules as the reference's: compute.glsl:225-330) runs over it instead:
//   * binned SAH over all three axes, one primitive per leaf;
//   * collapsed to 4 children per node (the child with the largest surface is opened first);
//   * child boxes quantised to 8 bits per plane relative to the node's own box and rounded
//     OUTWARD (checked with the exact arithmetic the traversal decodes with), so no hit is lost;
//   * the first kFastTopNodes nodes, which the kernels keep in LDS, are the root and, grown from it, the
//     nodes with the largest boxes (the ones a ray is most likely to visit); the rest breadth-first.
// Node record (16 words = 64 bytes; the reference's node is 64 bytes too but needs a divide per
// plane to decode, which is why the default mode expands it to 128 bytes on upload):
//   [0..2] origin xyz (f32)   [3] biased exponents ex | ey << 8 | ez << 16 (scale = 2^(e-127))
//   [4] lo_x[4] bytes  [5] hi_x[4]  [6] lo_y[4]  [7] hi_y[4]  [8] lo_z[4]  [9] hi_z[4]  [10..11] 0
//   [12..15] child words in the reference's metadata format (type << 30 | index; type 0 triangle,
//            1 sphere, 2 box: index into the uploaded buffer; type 3: node index), 0xFFFFFFFF = empty slot
//   plane = fma(code, scale, origin)
// The array ends with one extra all-empty node, which out-of-range indices are clamped to.

#include "rt_fastbvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace rt {

namespace {

struct Box3 {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) lo[a] = std::numeric_limits<float>::infinity(), hi[a] = -std::numeric_limits<float>::infinity();
    }
    void grow(const Box3 &b) {
        for (int a = 0; a < 3; a++) lo[a] = std::min(lo[a], b.lo[a]), hi[a] = std::max(hi[a], b.hi[a]);
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box3 box;
    float c[3];      // centroid
    uint32_t meta;   // type << 30 | index in the uploaded buffer
};

struct BinNode {
    Box3 box;
    uint32_t left = 0, right = 0;  // children (inner) ...
    uint32_t first = 0, count = 0; // ... or primitive range (leaf: count > 0)
};

inline float f(const uint32_t *w, int i) {
    float v;
    memcpy(&v, w + i, 4);
    return v;
}
inline uint32_t bits(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    return u;
}

// widens a box by a couple of units in the last place: the intersection routines accept hits a
// rounding error outside the exact primitive
void pad(Box3 &b) {
    for (int a = 0; a < 3; a++) {
        float m = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
        float e = m * 4.7683716e-7f + 1e-30f;  // 4 * 2^-23 relative
        b.lo[a] -= e;
        b.hi[a] += e;
    }
}

constexpr int kBins = 16;
constexpr uint32_t kMaxLeaf = 1;  // primitives per leaf

struct Builder {
    std::vector<Prim> prims;
    std::vector<BinNode> nodes;

    // Recursion depth: a SAH split may peel off one primitive at a time (geometrically spaced
    // centroids), which would make the depth grow with the primitive count; past kMaxSahDepth the
    // split is the median of the widest centroid axis, so the depth stays below kMaxSahDepth + log2(n).
    static constexpr int kMaxSahDepth = 48;
    uint32_t build(uint32_t first, uint32_t count, int depth = 0) {
        const uint32_t me = (uint32_t)nodes.size();
        nodes.emplace_back();
        Box3 box, cbox;
        box.reset();
        cbox.reset();
        for (uint32_t i = first; i < first + count; i++) {
            box.grow(prims[i].box);
            for (int a = 0; a < 3; a++) {
                cbox.lo[a] = std::min(cbox.lo[a], prims[i].c[a]);
                cbox.hi[a] = std::max(cbox.hi[a], prims[i].c[a]);
            }
        }
        nodes[me].box = box;
        if (count == 1) {
            nodes[me].first = first, nodes[me].count = 1;
            return me;
        }
        // best binned split over the three axes
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = 0;
        for (int a = 0; a < 3 && depth < kMaxSahDepth; a++) {
            const float ext = cbox.hi[a] - cbox.lo[a];
            if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
            Box3 bb[kBins];
            uint32_t bn[kBins] = {0};
            for (auto &b : bb) b.reset();
            const float k = (float)kBins / ext;
            for (uint32_t i = first; i < first + count; i++) {
                int bi = std::min(kBins - 1, std::max(0, (int)((prims[i].c[a] - cbox.lo[a]) * k)));
                bb[bi].grow(prims[i].box);
                bn[bi]++;
            }
            float right_area[kBins];
            uint32_t right_n[kBins];
            Box3 acc;
            acc.reset();
            uint32_t n = 0;
            for (int i = kBins - 1; i > 0; i--) {
                acc.grow(bb[i]);
                n += bn[i];
                right_area[i] = n ? acc.half_area() : 0.0f;
                right_n[i] = n;
            }
            acc.reset();
            n = 0;
            for (int i = 1; i < kBins; i++) {  // split between bin i-1 and bin i
                acc.grow(bb[i - 1]);
                n += bn[i - 1];
                if (n == 0 || right_n[i] == 0) continue;
                float cost = acc.half_area() * (float)n + right_area[i] * (float)right_n[i];
                if (cost < best_cost) best_cost = cost, best_axis = a, best_bin = i;
            }
        }
        const float area = box.half_area();
        const float split_cost = area > 0.0f ? 1.0f + best_cost / area : std::numeric_limits<float>::infinity();
        if (count <= kMaxLeaf && (best_axis < 0 || split_cost >= (float)count)) {
            nodes[me].first = first, nodes[me].count = count;
            return me;
        }
        uint32_t mid;
        if (best_axis >= 0) {
            const int a = best_axis;
            const float k = (float)kBins / (cbox.hi[a] - cbox.lo[a]), lo = cbox.lo[a];
            auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim &p) {
                int bi = std::min(kBins - 1, std::max(0, (int)((p.c[a] - lo) * k)));
                return bi < best_bin;
            });
            mid = (uint32_t)(it - prims.begin());
        } else {
            // no SAH split (all centroids coincide, or the depth limit): the median of the widest axis
            int a = 0;
            for (int k = 1; k < 3; k++)
                if (cbox.hi[k] - cbox.lo[k] > cbox.hi[a] - cbox.lo[a]) a = k;
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [a](const Prim &x, const Prim &y) { return x.c[a] < y.c[a]; });
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        const uint32_t l = build(first, mid - first, depth + 1);
        const uint32_t r = build(mid, first + count - mid, depth + 1);
        nodes[me].left = l, nodes[me].right = r;
        return me;
    }
};

// Smallest biased exponent e (scale = 2^(e-127)) and codes such that every child plane decodes
// outward of the exact one: lo' = fma(qlo, scale, origin) <= lo, hi' = fma(qhi, scale, origin) >= hi.
// Returns false if no exponent gives conservative codes (coordinates near the end of the float
// range, where the extent overflows): such a scene cannot be held by this node format.
bool quantise_axis(float origin, float extent, const float *clo, const float *chi, int nchild, uint32_t &e_out,
                   uint8_t *qlo, uint8_t *qhi) {
    int e = 1;
    if (!std::isfinite(extent) || !std::isfinite(origin)) return false;
    if (extent > 0.0f) {
        int ex;
        std::frexp(extent / 255.0f, &ex);  // extent/255 = m * 2^ex, m in [0.5, 1)  ->  2^ex >= extent/255
        e = std::min(254, std::max(1, ex + 127));
    }
    for (;; e++) {
        uint32_t sb = (uint32_t)e << 23;
        float scale;
        memcpy(&scale, &sb, 4);
        bool ok = true;
        for (int c = 0; c < nchild && ok; c++) {
            const double dlo = std::floor(((double)clo[c] - origin) / scale), dhi = std::ceil(((double)chi[c] - origin) / scale);
            int lo = (int)std::min(255.0, std::max(0.0, dlo)), hi = (int)std::min(255.0, std::max(0.0, dhi));
            while (lo > 0 && std::fma((float)lo, scale, origin) > clo[c]) lo--;
            while (hi < 255 && std::fma((float)hi, scale, origin) < chi[c]) hi++;
            if (std::fma((float)lo, scale, origin) > clo[c] || std::fma((float)hi, scale, origin) < chi[c]) ok = false;
            qlo[c] = (uint8_t)lo, qhi[c] = (uint8_t)hi;
        }
        if (ok) break;
        if (e >= 254) return false;
    }
    e_out = (uint32_t)e;
    return true;
}

}  // namespace

int build_fast_bvh(const uint32_t *tris, uint32_t n_tris, const uint32_t *spheres, uint32_t n_spheres,
                   const uint32_t *boxes, uint32_t n_boxes, FastBvh &out) {
    Builder b;
    b.prims.reserve((size_t)n_tris + n_spheres + n_boxes);
    auto add = [&](Box3 box, uint32_t meta) {
        for (int a = 0; a < 3; a++)
            if (!std::isfinite(box.lo[a]) || !std::isfinite(box.hi[a])) return;  // can never be hit by a finite ray test
        pad(box);
        Prim p;
        p.box = box;
        for (int a = 0; a < 3; a++) p.c[a] = 0.5f * box.lo[a] + 0.5f * box.hi[a];
        p.meta = meta;
        b.prims.push_back(p);
    };
    for (uint32_t i = 0; i < n_tris; i++) {
        const uint32_t *w = tris + (size_t)i * 16;
        Box3 box;
        for (int a = 0; a < 3; a++) {
            float v0 = f(w, a), v1 = f(w, 4 + a), v2 = f(w, 8 + a);
            box.lo[a] = std::min(v0, std::min(v1, v2));
            box.hi[a] = std::max(v0, std::max(v1, v2));
            if (v0 != v0 || v1 != v1 || v2 != v2) box.lo[a] = std::numeric_limits<float>::quiet_NaN();
        }
        add(box, (0u << 30) | i);
    }
    for (uint32_t i = 0; i < n_spheres; i++) {
        const uint32_t *w = spheres + (size_t)i * 8;
        const float r2 = f(w, 4);
        const float r = std::sqrt(std::max(r2, 0.0f)) * 1.000001f;
        Box3 box;
        for (int a = 0; a < 3; a++) box.lo[a] = f(w, a) - r, box.hi[a] = f(w, a) + r;
        if (!(r2 >= 0.0f)) box.lo[0] = std::numeric_limits<float>::quiet_NaN();
        add(box, (1u << 30) | i);
    }
    for (uint32_t i = 0; i < n_boxes; i++) {
        const uint32_t *w = boxes + (size_t)i * 8;
        Box3 box;
        for (int a = 0; a < 3; a++) {
            float p0 = f(w, a), p1 = f(w, 4 + a);
            box.lo[a] = std::min(p0, p1), box.hi[a] = std::max(p0, p1);
            if (p0 != p0 || p1 != p1) box.lo[a] = std::numeric_limits<float>::quiet_NaN();
        }
        add(box, (2u << 30) | i);
    }

    out = FastBvh();
    const uint32_t np = (uint32_t)b.prims.size();
    uint32_t empty[16] = {0};
    for (int k = 12; k < 16; k++) empty[k] = 0xFFFFFFFFu;
    empty[3] = 1u | (1u << 8) | (1u << 16);
    if (np == 0) {
        // a root with four empty children, and the sentinel
        out.nodes.insert(out.nodes.end(), empty, empty + 16);
        out.nodes.insert(out.nodes.end(), empty, empty + 16);
        out.n_nodes = 1;
        return 0;
    }
    b.nodes.reserve((size_t)np * 2);
    const uint32_t root = b.build(0, np);

    // collapse to 4 children per node, breadth-first numbering
    std::vector<uint32_t> queue;  // binary node that becomes wide node i
    queue.push_back(root);
    for (size_t head = 0; head < queue.size(); head++) {
        const BinNode &bn = b.nodes[queue[head]];
        uint32_t kids[4];
        int nk = 0;
        if (bn.count > 0) {
            kids[nk++] = queue[head];  // a single leaf as the whole tree
        } else {
            kids[nk++] = bn.left;
            kids[nk++] = bn.right;
            while (nk < 4) {
                int pick = -1;
                float pa = -1.0f;
                for (int k = 0; k < nk; k++) {
                    const BinNode &c = b.nodes[kids[k]];
                    if (c.count == 0 && c.box.half_area() > pa) pa = c.box.half_area(), pick = k;
                }
                if (pick < 0) break;
                const BinNode &c = b.nodes[kids[pick]];
                kids[pick] = c.left;
                kids[nk++] = c.right;
            }
        }
        uint32_t w[16] = {0};
        const Box3 &box = bn.box;
        float clo[3][4], chi[3][4];
        for (int k = 0; k < 4; k++) {
            if (k >= nk) {
                w[12 + k] = 0xFFFFFFFFu;
                continue;
            }
            const BinNode &c = b.nodes[kids[k]];
            for (int a = 0; a < 3; a++) clo[a][k] = c.box.lo[a], chi[a][k] = c.box.hi[a];
            if (c.count > 0) {
                w[12 + k] = b.prims[c.first].meta;  // one primitive per leaf
            } else {
                w[12 + k] = 0xC0000000u | (uint32_t)queue.size();
                queue.push_back(kids[k]);
            }
        }
        uint32_t ex[3];
        for (int a = 0; a < 3; a++) {
            uint8_t qlo[4] = {0, 0, 0, 0}, qhi[4] = {0, 0, 0, 0};
            if (!quantise_axis(box.lo[a], box.hi[a] - box.lo[a], clo[a], chi[a], nk, ex[a], qlo, qhi)) {
                out = FastBvh();
                return 1;
            }
            w[a] = bits(box.lo[a]);
            w[4 + 2 * a] = qlo[0] | (qlo[1] << 8) | (qlo[2] << 16) | ((uint32_t)qlo[3] << 24);
            w[5 + 2 * a] = qhi[0] | (qhi[1] << 8) | (qhi[2] << 16) | ((uint32_t)qhi[3] << 24);
        }
        w[3] = ex[0] | (ex[1] << 8) | (ex[2] << 16);
        out.nodes.insert(out.nodes.end(), w, w + 16);
    }
    out.n_nodes = (uint32_t)(out.nodes.size() / 16);
    {
        // Renumber: the nodes the kernels copy into LDS are chosen by box area from the root down
        // (always the frontier node with the largest box, ties to the smaller index), everything
        // else keeps its breadth-first order behind them.
        const uint32_t n = out.n_nodes;
        std::vector<uint32_t> order, rank(n, 0xFFFFFFFFu);
        order.reserve(n);
        auto emit = [&](uint32_t node) { rank[node] = (uint32_t)order.size(), order.push_back(node); };
        auto less = [](const std::pair<float, uint32_t> &x, const std::pair<float, uint32_t> &y) {
            return x.first < y.first || (x.first == y.first && x.second > y.second);
        };
        std::vector<std::pair<float, uint32_t>> heap;
        auto push_children = [&](uint32_t node) {
            for (int k = 0; k < 4; k++) {
                const uint32_t w = out.nodes[(size_t)node * 16 + 12 + k];
                if (w == 0xFFFFFFFFu || (w >> 30) != 3u) continue;
                const uint32_t child = w & 0x3FFFFFFFu;
                const float area = b.nodes[queue[child]].box.half_area();
                heap.emplace_back(area == area ? area : 0.0f, child);
                std::push_heap(heap.begin(), heap.end(), less);
            }
        };
        emit(0);
        push_children(0);
        while (!heap.empty() && order.size() < kFastTopNodes) {
            std::pop_heap(heap.begin(), heap.end(), less);
            const uint32_t node = heap.back().second;
            heap.pop_back();
            emit(node);
            push_children(node);
        }
        for (uint32_t i = 0; i < n; i++)
            if (rank[i] == 0xFFFFFFFFu) emit(i);
        std::vector<uint32_t> moved((size_t)n * 16);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t *src = &out.nodes[(size_t)order[i] * 16];
            uint32_t *dst = &moved[(size_t)i * 16];
            std::memcpy(dst, src, 64);
            for (int k = 12; k < 16; k++)
                if (dst[k] != 0xFFFFFFFFu && (dst[k] >> 30) == 3u) dst[k] = 0xC0000000u | rank[dst[k] & 0x3FFFFFFFu];
        }
        out.nodes.swap(moved);
    }
    out.nodes.insert(out.nodes.end(), empty, empty + 16);  // the sentinel
    return 0;
}

}  // namespace rt
