// Host-side builder of the fast-mode tree (no GPU dependency: also built into librt_scene.so).
#pragma once

#include <stdint.h>

#include <vector>

namespace rt {

// leading nodes of the fast-mode tree that the kernels keep in LDS (64 bytes each: 12 KB per workgroup)
constexpr uint32_t kFastTopNodes = 192;

struct FastBvh {
    std::vector<uint32_t> nodes;  // 16 words per node: n_nodes nodes plus one all-empty sentinel
    uint32_t n_nodes = 0;
};
// Builds the tree from host copies of the uploaded reference-layout buffers (record layouts:
// rt_fastbvh.cpp).  Returns 0, or 1 if the coordinates are so large that no conservative 8-bit
// child boxes exist (extents that overflow the float range); `out` is then empty.
int build_fast_bvh(const uint32_t *tris, uint32_t n_tris, const uint32_t *spheres, uint32_t n_spheres,
                    const uint32_t *boxes, uint32_t n_boxes, FastBvh &out);

}  // namespace rt
