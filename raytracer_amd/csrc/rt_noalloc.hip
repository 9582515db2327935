// rt_render_noalloc: one frame of the render path for callers that may not allocate (code under
// HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels as
// rt_render; everything those kernels need beyond the caller's inputs and outputs -- the device
// records derived from the reference-layout buffers and the ray counters -- lives in one
// file-scope device array of fixed size instead of buffers obtained from hipMalloc.
//
// The array exists once per GPU (each device that loads this code object holds its own copy at
// its own address) for the life of the process.  Host code never uses the array's name as an
// address: the device address is looked up once per device with hipGetSymbolAddress and that
// pointer is what kernels, memsets and copies receive.
//
// Calls on one GPU share that GPU's array: two calls that overlap in time on the same GPU (from
// different streams or threads) race.  Calls on one stream, or ordered by events, are fine.

#include "../../include/rt_mi355x.h"
#include "rt_internal.h"

#include <mutex>
#include <string>

namespace rt {

// ---- capacity of the scratch array (elements of each reference-layout buffer) ----------------
constexpr uint32_t kNaMaxNodes = RT_NOALLOC_MAX_NODES;
constexpr uint32_t kNaMaxTris = RT_NOALLOC_MAX_TRIANGLES;
constexpr uint32_t kNaMaxSpheres = RT_NOALLOC_MAX_SPHERES;
constexpr uint32_t kNaMaxBoxes = RT_NOALLOC_MAX_BOXES;
constexpr uint32_t kNaMaxPlanes = RT_NOALLOC_MAX_PLANES;
constexpr uint32_t kNaMaxMats = RT_NOALLOC_MAX_MATERIALS;

constexpr size_t align256(size_t n) { return (n + 255u) & ~(size_t)255u; }
// byte offsets of the regions; every derived array has one extra (sentinel) element
constexpr size_t kOffCounters = 0;
constexpr size_t kOffNodes = kOffCounters + align256(sizeof(Counters) * kCounterStripes);
constexpr size_t kOffTris = kOffNodes + align256(((size_t)kNaMaxNodes + 1) * kNodeVec * sizeof(float4));
constexpr size_t kOffSpheres = kOffTris + align256(((size_t)kNaMaxTris + 1) * kTriVec * sizeof(float4));
constexpr size_t kOffBoxes = kOffSpheres + align256(((size_t)kNaMaxSpheres + 1) * kSphereVec * sizeof(float4));
constexpr size_t kOffPlanes = kOffBoxes + align256(((size_t)kNaMaxBoxes + 1) * 2 * sizeof(float4));
constexpr size_t kOffMats = kOffPlanes + align256(((size_t)kNaMaxPlanes + 1) * 2 * sizeof(float4));
constexpr size_t kScratchBytes = kOffMats + align256(((size_t)kNaMaxMats + 1) * kMatVec * sizeof(float4));

}  // namespace rt

// about 8.7 MB per GPU
static __device__ __align__(256) unsigned char scratch[rt::kScratchBytes];

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_mu;
unsigned char *g_scratch_ptr[kMaxDevices];  // device address of `scratch` on each GPU, null until looked up

// Device address of this GPU's scratch array; `device` must be current.
hipError_t scratch_of(int device, unsigned char **out) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

}  // namespace

namespace rt {

// Looks the scratch address up ahead of time (called from rt_create with the device current), so
// that a first rt_render_noalloc under stream capture finds it ready.
hipError_t noalloc_prepare_device(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    unsigned char *p;
    return scratch_of(device, &p);
}

}  // namespace rt

#define NA_FAIL(msg)                                 \
    do {                                             \
        rt::set_last_error(std::string("rt_render_noalloc: ") + (msg)); \
        return 1;                                    \
    } while (0)
#define NA_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) NA_FAIL(std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

extern "C" int rt_render_noalloc(int device, const rt_scene_view *sv, int width, int height, uint32_t frame,
                                 float bounces, float *sample_rgba, float *acc_rgba, uint32_t accumulated_frames,
                                 void *stream) {
    using namespace rt;
    if (!sv) NA_FAIL("scene view is null");
    if (!sample_rgba) NA_FAIL("sample_rgba is null");
    if (width <= 0 || height <= 0) NA_FAIL("size must be positive");
    if (!(bounces >= 0.0f)) NA_FAIL("bounces must not be negative");
    if ((size_t)width * (size_t)height > ((size_t)1 << 28)) NA_FAIL("image too large");
    if (device < 0 || device >= kMaxDevices) NA_FAIL("no such HIP device");
    struct {
        const char *name;
        const void *ptr;
        uint32_t n, cap;
    } in[] = {{"bvh", sv->bvh, sv->n_nodes, kNaMaxNodes},
              {"triangles", sv->triangles, sv->n_triangles, kNaMaxTris},
              {"spheres", sv->spheres, sv->n_spheres, kNaMaxSpheres},
              {"boxes", sv->boxes, sv->n_boxes, kNaMaxBoxes},
              {"planes", sv->planes, sv->n_planes, kNaMaxPlanes},
              {"materials", sv->materials, sv->n_materials, kNaMaxMats}};
    for (const auto &b : in) {
        if (b.n > b.cap)
            NA_FAIL(std::string(b.name) + ": " + std::to_string(b.n) + " elements exceed the fixed capacity of " +
                    std::to_string(b.cap) + " (use rt_upload + rt_render for larger scenes)");
        if (b.n && !b.ptr) NA_FAIL(std::string(b.name) + ": null pointer with a non-zero count");
    }
    int ndev = 0;
    NA_HIP(hipGetDeviceCount(&ndev));
    if (device >= ndev) NA_FAIL("no such HIP device");
    NA_HIP(hipSetDevice(device));
    unsigned char *base = nullptr;
    NA_HIP(scratch_of(device, &base));
    hipStream_t s = static_cast<hipStream_t>(stream);

    Counters *counters = reinterpret_cast<Counters *>(base + kOffCounters);
    float4 *nodes = reinterpret_cast<float4 *>(base + kOffNodes);
    float4 *tris = reinterpret_cast<float4 *>(base + kOffTris);
    float4 *spheres = reinterpret_cast<float4 *>(base + kOffSpheres);
    float4 *boxes = reinterpret_cast<float4 *>(base + kOffBoxes);
    float4 *planes = reinterpret_cast<float4 *>(base + kOffPlanes);
    float4 *mats = reinterpret_cast<float4 *>(base + kOffMats);

    // the array keeps the previous call's contents: the counters must start at zero; the record
    // regions are rewritten for elements [0, n] (n = the zero-source sentinel), and nothing reads
    // beyond n, so a smaller scene after a larger one sees none of the stale tail
    NA_HIP(hipMemsetAsync(counters, 0, sizeof(Counters) * kCounterStripes, s));
    // the leaf regions follow the nodes in the same array: positions in units of 32 bytes from the first node
    const LeafRegions leaves = {(uint32_t)((kOffTris - kOffNodes) / 32), (uint32_t)((kOffSpheres - kOffNodes) / 32),
                                (uint32_t)((kOffBoxes - kOffNodes) / 32), sv->n_triangles, sv->n_spheres, sv->n_boxes};
    launch_repack_nodes(static_cast<const uint32_t *>(sv->bvh), sv->n_nodes, sv->scene_min, sv->scene_extent, nullptr,
                        nullptr, leaves, nodes, s);
    launch_repack_tris(static_cast<const uint32_t *>(sv->triangles), sv->n_triangles, tris, s);
    launch_repack_spheres(static_cast<const uint32_t *>(sv->spheres), sv->n_spheres, spheres, s);
    launch_copy_raw32(static_cast<const uint32_t *>(sv->boxes), sv->n_boxes, boxes, s);
    launch_copy_raw32(static_cast<const uint32_t *>(sv->planes), sv->n_planes, planes, s);
    launch_repack_mats(static_cast<const uint32_t *>(sv->materials), sv->n_materials, mats, s);

    TraceParams p;
    p.scene.nodes = nodes;
    p.scene.tris = tris;
    p.scene.spheres = spheres;
    p.scene.boxes = boxes;
    p.scene.planes = planes;
    p.scene.mats = mats;
    p.scene.n_nodes = sv->n_nodes;
    p.scene.n_tris = sv->n_triangles;
    p.scene.n_spheres = sv->n_spheres;
    p.scene.n_boxes = sv->n_boxes;
    p.scene.n_planes = sv->n_planes;
    p.scene.n_mats = sv->n_materials;
    p.scene.fnodes = nullptr;  // fast mode is a context feature; this entry point renders the reference's tree
    p.scene.n_fnodes = 0;
    p.scene.unit_tris = leaves.unit_tris, p.scene.unit_spheres = leaves.unit_spheres, p.scene.unit_boxes = leaves.unit_boxes;
    for (int i = 0; i < 16; i++) p.scene.inv_vp[i] = sv->inv_view_proj[i];
    for (int i = 0; i < 3; i++) {
        p.scene.eye[i] = sv->eye[i];
        p.scene.smin[i] = sv->scene_min[i];
        p.scene.sext[i] = sv->scene_extent[i];
    }
    p.W = width;
    p.H = height;
    p.WW = (width / 8) * 8;
    p.HH = (height / 4) * 4;
    p.time = (float)frame;
    p.bounces = bounces;
    p.sample = reinterpret_cast<float4 *>(sample_rgba);
    p.sum = nullptr;
    p.aux = nullptr;
    p.order = nullptr;  // macro tiles in index order: an ordering by cost would need buffers of its own
    p.cost = nullptr;
    p.counters = counters;
    p.shard = 0;
    p.nshards = 1;
    p.tiles_x = (uint32_t)((p.WW + 15) / 16);
    p.tiles_y = (uint32_t)((p.HH + 15) / 16);
    p.ntiles = p.tiles_x * p.tiles_y;
    // scenes within this entry point's limits are cache-resident: the variant with leaf steps of
    // their own is the faster one there (it cannot be timed here: that would need a host wait)
    launch_trace(p, 0, false, false, true, s);
    if (acc_rgba)
        launch_accumulate(p.sample, reinterpret_cast<float4 *>(acc_rgba), (size_t)width * height,
                          (float)accumulated_frames, s);
    NA_HIP(hipGetLastError());
    return 0;
}
