// Internal interface between the C ABI (rt_api.cpp) and the device code (rt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "rt_fastbvh.h"

struct rt_ctx;

namespace rt {

// ---- device-side record layouts (what the repack kernels produce) -------------------------
// Node: 128 B = 8 x float4, one L2 cache line, child boxes already decoded to fp32:
//   [0] min_x[4] [1] max_x[4] [2] min_y[4] [3] max_y[4] [4] min_z[4] [5] max_z[4]
//   [6] metadata[4] (bit pattern, type<<30 | index)   [7] unused
// Triangle: 64 B = 4 x float4:
//   [0] v0.xyz, 0   [1] e1.xyz, 0   [2] e2.xyz, 0   [3] normal.xyz, material
//   (the test reads 48 B, an accepted hit one 16-byte record more)
// Sphere: 32 B = 2 x float4:   [0] centre.xyz, radius^2   [1] material, 0, 0, 0
// Material: 64 B = 4 x float4:
//   [0] emitted.rgb (emission*strength), smoothness   [1] specular tint.rgb, transmission
//   [2] diffuse albedo.rgb (colour*(1-metallic)), f0  [3] ior, ggx a2, metallic, 0
// Boxes and planes keep the reference layout (2 x float4 each).
#ifndef RT_LDS_TOP
#define RT_LDS_TOP 109     // leading node records (the top of the tree) a workgroup of the megakernel keeps in LDS: what
                           // fits in the 20 KB a workgroup may use while 8 of them share a CU; 0 = none
#endif
constexpr uint32_t kLdsTopNodes = RT_LDS_TOP;
constexpr int kNodeVec = 8;
constexpr int kTriVec = 4;
constexpr int kSphereVec = 2;
constexpr int kMatVec = 4;

struct DevScene {
    // nodes, tris, spheres and boxes are regions of ONE allocation that begins with the nodes, and a
    // child word in a device node record names its record by position: type << 30 | offset from
    // `nodes` in units of 32 bytes (unit_* = where the regions begin).  A traversal step thus forms
    // any record's address with one shift-and-add, whatever the kind; reference_word() in
    // rt_kernels.hip turns such a word back into the reference's type << 30 | index.
    const float4 *nodes;
    const float4 *tris;
    const float4 *spheres;
    const float4 *boxes;
    const float4 *planes;
    const float4 *mats;
    // element counts; every repacked array carries one extra all-zero-source element at index n,
    // which is what an out-of-range index reads (GL robust buffer access returns zeros)
    uint32_t n_nodes, n_tris, n_spheres, n_boxes, n_planes, n_mats;
    uint32_t unit_tris, unit_spheres, unit_boxes;
    // fast mode only (rt_fastbvh.cpp): the library's own tree, 4 x float4 per node, n_fnodes nodes + a sentinel
    const float4 *fnodes;
    uint32_t n_fnodes;
    float inv_vp[16];
    float eye[3];
    float smin[3], sext[3];  // scene_min / scene_extent uniforms (child boxes are decoded with them)
};

// Device-side counters.  To keep same-address atomics rare, the kernels spread their updates over
// kCounterStripes copies (one 128-byte line each, chosen by workgroup index); the host adds them up.
constexpr int kCounterStripes = 64;
struct Counters {  // 16 x u64 = one 128-byte line
    unsigned long long rays, pixel_samples, nodes, tri_tests, sphere_tests, box_tests, max_stack, stack_overflows;
    unsigned long long hits;
    // Clock sample of the megakernel: a few of its wavefronts (one workgroup in 61) add the shader
    // cycles they lived for (s_memtime) and the same span in ticks of the constant 100 MHz counter
    // (s_memrealtime); cycles / ticks * 0.1 is the shader clock in GHz those wavefronts ran at.
    unsigned long long clk_cycles, clk_ticks;
    unsigned long long pad[5];
};

struct TraceParams {
    DevScene scene;
    int W, H;            // image size
    int WW, HH;          // pixels the reference dispatch covers: 8*(W/8) x 4*(H/4)
    float time;          // the `time` uniform (frame index)
    float bounces;       // the `bounces` uniform
    float4 *sample;      // mode 0: sRGB-encoded sample image (may be null)
    float4 *sum;         // mode 1: linear radiance sum image
    uint32_t *aux;       // 16 words per pixel or null
    Counters *counters;
    uint32_t shard, nshards;  // mode 1 tile ownership
    uint32_t tiles_x, tiles_y, ntiles;  // 16x16 pixel tiles
    // Order of the macro tiles within each XCD (kLptPerXcd entries per XCD) and where the wavefronts
    // add the cycles they took, per macro tile; both null: macro tiles in index order, nothing recorded.
    const uint32_t *order;
    uint32_t *cost;
};
#ifndef RT_MACRO
#define RT_MACRO 2  // a macro tile is RT_MACRO x RT_MACRO tiles of 16x16 pixels; the host counts them too
#endif
constexpr uint32_t kLptPerXcd = 1024;  // macro tiles per XCD the ordering can hold
// sorts every XCD's macro tiles by the cost recorded for them (most expensive first) and clears the costs
void launch_lpt_sort(uint32_t *cost, uint32_t *order, uint32_t n_macro, hipStream_t s);

// ---- wavefront pipeline: ray queue in device memory ----------------------------------------
struct WaveQueues {
    float4 *rays[2];  // [0]: ray records, 2 x float4 per slot ([1] unused)
    float2 *hits;     // per slot: distance, primitive metadata word
    float4 *paths;    // per slot: 2 x float4 (throughput+rng, light+ior)
};

// ---- fast mode: the library's own acceleration structure (rt_fastbvh.h / rt_fastbvh.cpp) -----
// one frame with the fast-mode traversal (same modes and flags as launch_trace)
void launch_trace_fast(const TraceParams &p, int mode, bool aux, bool stats, hipStream_t s);
// one frame with the regrouping megakernel (RT_PIPELINE=regroup; same results as launch_trace)
void launch_trace_regroup(const TraceParams &p, int mode, bool aux, bool stats, hipStream_t s);

// repack kernels: reference byte layouts -> device records (n elements + 1 zero-source sentinel)
void launch_reciprocal_selftest(unsigned long long *mismatches, hipStream_t s);
// Where the leaf records lie relative to the node records (units of 32 bytes) and how many there are:
// the node repack turns every child word into type << 30 | position (see DevScene).
struct LeafRegions {
    uint32_t unit_tris, unit_spheres, unit_boxes;
    uint32_t n_tris, n_spheres, n_boxes;
};
void launch_repack_nodes(const uint32_t *src, uint32_t n, const float smin[3], const float sext[3],
                         const uint32_t *order, const uint32_t *rank, const LeafRegions &leaves, float4 *dst,
                         hipStream_t s);
void launch_repack_tris(const uint32_t *src, uint32_t n, float4 *dst, hipStream_t s);
void launch_repack_spheres(const uint32_t *src, uint32_t n, float4 *dst, hipStream_t s);
void launch_repack_mats(const uint32_t *src, uint32_t n, float4 *dst, hipStream_t s);
void launch_copy_raw32(const uint32_t *src, uint32_t n, float4 *dst, hipStream_t s);  // boxes/planes + sentinel

// mode 0 = reference frame (writes sample), mode 1 = add linear radiance into sum
// leaf_steps: the variant with leaf steps of their own instead of the unified node / leaf step
void launch_trace(const TraceParams &p, int mode, bool aux, bool stats, bool leaf_steps, hipStream_t s);
// one frame through the wavefront pipeline: generate, then per bounce extend + shade
void launch_wavefront_frame(const TraceParams &p, const WaveQueues &q, int mode, bool aux, bool stats,
                            uint32_t extend_waves, uint32_t refill_threshold, bool bin_by_octant, hipStream_t s);
// accumulate pass: acc = srgb(lin(acc) + (lin(sample) - lin(acc)) / (frames + 1))
void launch_accumulate(const float4 *sample, float4 *acc, size_t npixels, float frames, hipStream_t s);
// acc = srgb(sum / spp)
void launch_fold_sum(float4 *target, float4 *part, size_t npixels, hipStream_t s);
void launch_finalize_sum(const float4 *sum, float4 *acc, size_t npixels, uint32_t spp, hipStream_t s);

// ---- what rt_comm.cpp needs from a context (rt_api.cpp) -------------------------------------
int ctx_device(rt_ctx *c);
void *ctx_comm(rt_ctx *c);                     // the RCCL communicator, or null
void ctx_set_comm(rt_ctx *c, void *comm, int rank, int world);
double *ctx_comm_scratch(rt_ctx *c);           // device buffer of RT_COMM_MAX_SCALARS doubles
// joins the trace streams (folds pending sums) and hands out the main stream, the context's own
// sum image and its size in floats
int ctx_join_for_collective(rt_ctx *c, hipStream_t *stream, float **own_sum, size_t *nfloats);
void comm_destroy(void *comm);                 // rt_comm.cpp

// message rt_last_error() returns on the calling thread (rt_api.cpp)
void set_last_error(const std::string &message);
// rt_noalloc.hip: resolves the device address of the fixed scratch array of GPU `device` (which
// must be current) so that a later rt_render_noalloc finds it ready
hipError_t noalloc_prepare_device(int device);

}  // namespace rt
