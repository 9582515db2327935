// Device code for gfx950 (MI355X): buffer repacking, the path-tracing kernel, accumulation.
//
// Numerics contract.  The reference's results are defined by how its GLSL is evaluated:
// every + - * / separately rounded binary32, no contraction, except inside sin/cos/exp2/log2
// where specific fused multiply-adds are part of the algorithm.  This file must therefore be
// compiled with -ffp-contract=off and with correctly rounded fp32 divide/sqrt (hipcc's default);
// fused operations are written explicitly as __builtin_fmaf.  Expression association below is
// deliberate everywhere; do not "simplify" it.
//
// Reference: shaders/compute.glsl (main :451-688 and callees).

#include "rt_internal.h"

namespace rt {

#define RT_BLOCK 256
// RT_LDS_TOP (rt_internal.h): number of leading device node records kept in LDS per workgroup
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 4   // megakernel: wavefronts per workgroup, each an 8x8 quarter of a 16x16 tile (the regrouping kernel assumes 4)
#endif
#define RT_TRACE_BLOCK (64 * RT_TRACE_WAVES)
#ifndef RT_TOP_STRIDE
#define RT_TOP_STRIDE 7    // float4 per cached node: 112 B; 7 is odd, so 16 lanes on 16 consecutive nodes use 16 different bank groups
#endif
// Traversal stack: the first RT_STACK_LDS entries of every lane live in LDS (lane-contiguous, so
// conflict-free); deeper entries spill to per-lane private memory.  The reference's own array
// has 32 entries and no overflow guard; here pushes beyond RT_STACK_LDS + RT_STACK_SPILL are
// dropped and counted.
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 8
#endif
#ifndef RT_STACK_SPILL
#define RT_STACK_SPILL 56
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 8  // megakernel: waves per SIMD the register allocator must leave room for
#endif
#define RT_FAST_TOP 192    // fast mode: leading nodes (64 bytes each) kept in LDS per workgroup: 12 KB
static_assert(RT_FAST_TOP == kFastTopNodes, "the builder (rt_fastbvh.cpp) puts exactly the nodes of the LDS copy first");
// RT_MACRO (rt_internal.h): megakernel tile placement, macro tile = RT_MACRO x RT_MACRO tiles of 16x16 pixels, dealt to the XCDs in turn
#ifndef RT_EXT_STACK_LDS
#define RT_EXT_STACK_LDS 8   // traversal kernel of the wavefront pipeline: 2 KB of LDS stack per wavefront
#endif
#ifndef RT_EXTEND_MIN_WAVES
#define RT_EXTEND_MIN_WAVES 8
#endif

// ------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

struct v3 {
    float x, y, z;
};
__device__ __forceinline__ v3 V(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ v3 operator-(v3 a) { return V(-a.x, -a.y, -a.z); }
// dot(vec3) association: x + (y + z)
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ v3 normalize(v3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
__device__ __forceinline__ v3 reflect(v3 I, v3 N) { return I - N * (2.0f * dot(N, I)); }

// ------------------------------------------------------------------------------------------
// transcendental functions in the exact polynomial forms the reference's results come from
// (they are not libm: see compute.glsl's use of pow/sin/cos at :358-361,:396-402,:553,:697,:705)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float d_exp2(float x) {
    x = fminf(fmaxf(x, -127.0f), 128.0f);
    float ip = floorf(x);
    float fp = x - ip;
    float x2 = fp * fp;
    // even/odd split Horner, degree 5
    float even = 0.00898934009049466391101f;
    even = __builtin_fmaf(x2, even, 0.240153617044375388211f);
    even = __builtin_fmaf(x2, even, 1.0f);
    float odd = 0.00187757667519147912699f;
    odd = __builtin_fmaf(x2, odd, 0.0558263180532956664775f);
    odd = __builtin_fmaf(x2, odd, 0.693153073200168932794f);
    float p = __builtin_fmaf(odd, fp, even);
    uint32_t bits = (uint32_t)((int)ip + 127) << 23;
    return u2f(bits) * p;
}

__device__ __forceinline__ float d_log2(float x) {
    uint32_t i = f2u(x);
    float e = (float)((int)((i >> 23) & 0xffu) - 127);
    float m = u2f((i & 0x007fffffu) | 0x3f800000u);
    float y = (m - 1.0f) / (m + 1.0f);
    float z = y * y;
    float z2 = z * z;
    // even/odd split Horner, degree 4 in z
    float even = 0.406718052498846252698f;
    even = __builtin_fmaf(z2, even, 0.577440339438736392009f);
    even = __builtin_fmaf(z2, even, 2.88539009343309178325f);
    float odd = 0.403343858251329912514f;
    odd = __builtin_fmaf(z2, odd, 0.961791550404184197881f);
    float p = __builtin_fmaf(odd, z, even);
    float r = __builtin_fmaf(y, p, e);
    if (x >= __builtin_inff()) r = __builtin_inff();
    if (x == 0.0f) r = -__builtin_inff();
    if (x < 0.0f) r = __builtin_nanf("");
    return r;
}

__device__ __forceinline__ float d_pow(float x, float y) { return d_exp2(d_log2(x) * y); }

// sin and cos of the same angle (Cephes-style range reduction, both polynomials shared)
__device__ __forceinline__ void d_sincos(float a, float &s_out, float &c_out) {
    float x = fabsf(a);
    int j = (int)(x * 1.27323954473516f);
    j = (j + 1) & ~1;
    float y = (float)j;
    x = __builtin_fmaf(y, -0.78515625f, x);
    x = __builtin_fmaf(y, -2.4187564849853515625e-4f, x);
    x = __builtin_fmaf(y, -3.77489497744594108e-8f, x);
    float z = x * x;
    float yc = 2.443315711809948e-5f;
    yc = __builtin_fmaf(yc, z, -1.388731625493765e-3f);
    yc = __builtin_fmaf(yc, z, 4.166664568298827e-2f);
    yc = yc * z;
    yc = yc * z;
    yc = __builtin_fmaf(z, -0.5f, yc);
    yc = yc + 1.0f;
    float ys = -1.9515295891e-4f;
    ys = __builtin_fmaf(ys, z, 8.3321608736e-3f);
    ys = __builtin_fmaf(ys, z, -1.6666654611e-1f);
    ys = ys * z;
    ys = __builtin_fmaf(ys, x, x);
    uint32_t sign_s = (f2u(a) & 0x80000000u) ^ (((uint32_t)j & 4u) << 29);
    bool sin_uses_sinpoly = ((j & 2) == 0);
    int j2 = j - 2;
    uint32_t sign_c = ((~(uint32_t)j2) & 4u) << 29;
    bool cos_uses_sinpoly = ((j2 & 2) == 0);
    s_out = u2f(f2u(sin_uses_sinpoly ? ys : yc) ^ sign_s);
    c_out = u2f(f2u(cos_uses_sinpoly ? ys : yc) ^ sign_c);
}

// compute.glsl:701-706 and :693-698
__device__ __forceinline__ float s2l(float c) {
    if (c <= 0.04045f) return c / 12.92f;
    return d_pow((c + 0.055f) / (1.0f + 0.055f), 2.4f);
}
__device__ __forceinline__ float l2s(float c) {
    if (c <= 0.0031308f) return 12.92f * c;
    return (1.0f + 0.055f) * d_pow(c, 1.0f / 2.4f) - 0.055f;
}

// fast mode only: reciprocal of a direction component, kept finite
__device__ __forceinline__ float fast_rcp(float d) { return fabsf(d) > 1e-20f ? 1.0f / d : copysignf(1e20f, d); }

// compute.glsl:749-760
__device__ __forceinline__ float draw(uint32_t &s) {
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    return (float)s * (1.0f / 4294967296.0f);
}

// ------------------------------------------------------------------------------------------
// repack kernels (run once per upload): reference layouts -> cache-line-aligned device records.
// Every derived value is computed with exactly the operations the shader would apply per use,
// so hoisting it here cannot change a result bit.
// ------------------------------------------------------------------------------------------
// BVH node, decoded once here, compute.glsl:244-261: f = float(q)/65535.0 ; coord = scene_min + f*scene_extent
// Device record i is built from source node order[i]; child indices are rewritten with
// rank[] (rank[order[i]] = i).  The renumbering (node_order in rt_api.cpp) puts the nodes a ray is
// most likely to visit -- the root and, grown from it, the nodes with the largest boxes -- first,
// where the kernels' LDS copy takes them from, and the rest in breadth-first order; nothing outside
// the traversal ever sees a node index, so results are unchanged.  Indices that point outside the buffer stay as
// they are (they select the all-zero record either way).  order == rank == nullptr keeps the
// source numbering.
__global__ void repack_nodes_kernel(const uint32_t *__restrict__ src, uint32_t n, float3 smin, float3 sext,
                                    const uint32_t *__restrict__ order, const uint32_t *__restrict__ rank,
                                    LeafRegions lr, float4 *__restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t w[16];
    const uint32_t from = (i < n) ? (order ? order[i] : i) : 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = (i < n) ? src[(size_t)from * 16 + k] : 0u;
#pragma unroll
    for (int k = 12; k < 16; k++) {
        // The child word becomes type << 30 | the record's position, in units of 32 bytes from the first
        // node record (DevScene).  An index beyond its array -- the reference's empty slot is node
        // 0x3FFFFFFF -- names the all-zero record that ends every array, which is what such a read
        // delivers in the reference (robust buffer access).  No word is 0xFFFFFFFF afterwards.
        const uint32_t type = w[k] >> 30, idx = w[k] & 0x3FFFFFFFu;
        uint32_t unit;
        if (type == 3u) unit = 4u * (idx < n ? (rank ? rank[idx] : idx) : n);
        else if (type == 0u) unit = lr.unit_tris + 2u * min(idx, lr.n_tris);
        else if (type == 1u) unit = lr.unit_spheres + min(idx, lr.n_spheres);
        else unit = lr.unit_boxes + min(idx, lr.n_boxes);
        w[k] = (type << 30) | unit;
    }
    const float mn[3] = {smin.x, smin.y, smin.z}, ex[3] = {sext.x, sext.y, sext.z};
    float lo[3][4], hi[3][4];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float l = (float)(w[4 * a + c] & 0xFFFFu) / 65535.0f;
            float h = (float)(w[4 * a + c] >> 16) / 65535.0f;
            lo[a][c] = mn[a] + l * ex[a];
            hi[a][c] = mn[a] + h * ex[a];
        }
    float4 *o = dst + (size_t)i * kNodeVec;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        o[2 * a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
        o[2 * a + 1] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
    }
    o[6] = make_float4(u2f(w[12]), u2f(w[13]), u2f(w[14]), u2f(w[15]));
    o[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Triangle, compute.glsl:880-881: edge1 = v1 - v0, edge2 = v2 - v0
__global__ void repack_tris_kernel(const uint32_t *__restrict__ src, uint32_t n, float4 *__restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = (i < n) ? src[(size_t)i * 16 + k] : 0u;
    v3 v0 = V(u2f(w[0]), u2f(w[1]), u2f(w[2])), v1 = V(u2f(w[4]), u2f(w[5]), u2f(w[6])),
       v2 = V(u2f(w[8]), u2f(w[9]), u2f(w[10]));
    v3 e1 = v1 - v0, e2 = v2 - v0;
    float4 *o = dst + (size_t)i * kTriVec;
    // the intersection test reads records 0..2, the accepted hit only record 3
    o[0] = make_float4(v0.x, v0.y, v0.z, 0.f);
    o[1] = make_float4(e1.x, e1.y, e1.z, 0.f);
    o[2] = make_float4(e2.x, e2.y, e2.z, 0.f);
    o[3] = make_float4(u2f(w[12]), u2f(w[13]), u2f(w[14]), u2f(w[3]));
}

__global__ void repack_spheres_kernel(const uint32_t *__restrict__ src, uint32_t n, float4 *__restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = (i < n) ? src[(size_t)i * 8 + k] : 0u;
    dst[(size_t)i * 2] = make_float4(u2f(w[0]), u2f(w[1]), u2f(w[2]), u2f(w[4]));
    dst[(size_t)i * 2 + 1] = make_float4(u2f(w[5]), 0.f, 0.f, 0.f);
}

__global__ void copy_raw32_kernel(const uint32_t *__restrict__ src, uint32_t n, float4 *__restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = (i < n) ? src[(size_t)i * 8 + k] : 0u;
    dst[(size_t)i * 2] = make_float4(u2f(w[0]), u2f(w[1]), u2f(w[2]), u2f(w[3]));
    dst[(size_t)i * 2 + 1] = make_float4(u2f(w[4]), u2f(w[5]), u2f(w[6]), u2f(w[7]));
}

// Material, compute.glsl:511-549.  The two mix(.., .., metallic) calls are evaluated in the
// x*(1-t) + y*t form sharing (1 - metallic); pow(x, 2.0) is x*x.
__global__ void repack_mats_kernel(const uint32_t *__restrict__ src, uint32_t n, float4 *__restrict__ dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t c = (i < n) ? src[(size_t)i * 3] : 0u, e = (i < n) ? src[(size_t)i * 3 + 1] : 0u,
             p = (i < n) ? src[(size_t)i * 3 + 2] : 0u;
    v3 col = V(s2l((float)((c >> 21) & 0x7FFu) / 2047.0f), s2l((float)((c >> 10) & 0x7FFu) / 2047.0f),
               s2l((float)(c & 0x3FFu) / 1023.0f));
    float smooth = (float)(p & 0xFFu) / 255.0f;
    float metal = (float)((p >> 8) & 0xFFu) / 255.0f;
    float trans = (float)((p >> 16) & 0xFFu) / 255.0f;
    float ior = ((float)((p >> 24) & 0xFFu) / 255.0f) * 3.0f;
    v3 em = V(s2l((float)(e & 0xFFu) / 255.0f), s2l((float)((e >> 8) & 0xFFu) / 255.0f),
              s2l((float)((e >> 16) & 0xFFu) / 255.0f));
    float strength = (float)((e >> 24) & 0xFFu);
    v3 emitted = em * strength;
    float omm = 1.0f - metal;
    v3 tint = V(omm + col.x * metal, omm + col.y * metal, omm + col.z * metal);
    float q = (1.0f - ior) / (1.0f + ior);
    float f0d = q * q;
    float f0 = f0d * omm + metal;
    v3 albedo = col * omm;
    float rough = 1.0f - smooth;
    float a = rough * rough;
    float a2 = a * a;
    float4 *o = dst + (size_t)i * kMatVec;
    o[0] = make_float4(emitted.x, emitted.y, emitted.z, smooth);
    o[1] = make_float4(tint.x, tint.y, tint.z, trans);
    o[2] = make_float4(albedo.x, albedo.y, albedo.z, f0);
    o[3] = make_float4(ior, a2, metal, 0.f);
}

static inline unsigned grid_for(uint32_t n_plus_sentinel) { return (n_plus_sentinel + 255u) / 256u; }

void launch_repack_nodes(const uint32_t *src, uint32_t n, const float smin[3], const float sext[3],
                         const uint32_t *order, const uint32_t *rank, const LeafRegions &leaves, float4 *dst,
                         hipStream_t s) {
    hipLaunchKernelGGL(repack_nodes_kernel, dim3(grid_for(n + 1)), dim3(256), 0, s, src, n,
                       make_float3(smin[0], smin[1], smin[2]), make_float3(sext[0], sext[1], sext[2]), order, rank,
                       leaves, dst);
}
void launch_repack_tris(const uint32_t *src, uint32_t n, float4 *dst, hipStream_t s) {
    hipLaunchKernelGGL(repack_tris_kernel, dim3(grid_for(n + 1)), dim3(256), 0, s, src, n, dst);
}
void launch_repack_spheres(const uint32_t *src, uint32_t n, float4 *dst, hipStream_t s) {
    hipLaunchKernelGGL(repack_spheres_kernel, dim3(grid_for(n + 1)), dim3(256), 0, s, src, n, dst);
}
void launch_repack_mats(const uint32_t *src, uint32_t n, float4 *dst, hipStream_t s) {
    hipLaunchKernelGGL(repack_mats_kernel, dim3(grid_for(n + 1)), dim3(256), 0, s, src, n, dst);
}
void launch_copy_raw32(const uint32_t *src, uint32_t n, float4 *dst, hipStream_t s) {
    hipLaunchKernelGGL(copy_raw32_kernel, dim3(grid_for(n + 1)), dim3(256), 0, s, src, n, dst);
}

// ------------------------------------------------------------------------------------------
// traversal (compute.glsl:225-330)
// ------------------------------------------------------------------------------------------
struct LaneStats {
    uint32_t nodes, tri, sph, box, max_sp, overflow, hits;
};

// compute.glsl:875-919 up to the distance; normal and position are derived after traversal
// 1.0f / x, correctly rounded.  For 1e-6 <= |x| <= 1e30 the hardware's reciprocal estimate with one
// refinement step, fma(fma(-x, r, 1), r, r), IS the correctly rounded quotient -- checked over every
// float in that range on the device itself (reciprocal_selftest_kernel below, rt_selftest in the
// ABI) -- and costs 3 instructions instead of the 10 of the general division, which anything outside
// the range (infinities and NaNs included) still takes.
__device__ __forceinline__ float reciprocal(float x) {
#ifndef RT_GENERAL_DIVISION  // (A/B builds)
    if (__builtin_expect(fabsf(x) <= 1e30f && fabsf(x) >= 1e-6f, 1)) {
        const float r = __builtin_amdgcn_rcpf(x);
        return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    }
#endif
    return 1.0f / x;
}

// Counts the floats for which reciprocal() differs from the division, over all 2^32 bit patterns.
__global__ void reciprocal_selftest_kernel(unsigned long long *mismatches) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint64_t u = tid; u < (1ull << 32); u += stride) {
        const float x = u2f((uint32_t)u);
        const float want = 1.0f / x, got = reciprocal(x);
        if (f2u(want) != f2u(got) && !(want != want && got != got)) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}
void launch_reciprocal_selftest(unsigned long long *mismatches, hipStream_t s) {
    hipLaunchKernelGGL(reciprocal_selftest_kernel, dim3(4096), dim3(256), 0, s, mismatches);
}

__device__ __forceinline__ bool tri_distance(v3 O, v3 D, float4 r0, float4 r1, float4 r2, float &t_out) {
    v3 v0 = V(r0.x, r0.y, r0.z), e1 = V(r1.x, r1.y, r1.z), e2 = V(r2.x, r2.y, r2.z);
    v3 p = cross(D, e2);
    float det = dot(e1, p);
    if (fabsf(det) < 1e-6f) return false;
    float inv = reciprocal(det);
    v3 tv = O - v0;
    float u = dot(tv, p) * inv;
    if (u < 0.0f || u > 1.0f) return false;
    v3 q = cross(tv, e1);
    float v = dot(D, q) * inv;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(e2, q) * inv;
    if (t <= 0.0f) return false;
    t_out = t;
    return true;
}

// compute.glsl:808-836
__device__ __forceinline__ bool sphere_distance(v3 O, v3 D, float4 s0, float &t_out) {
    v3 L = V(s0.x, s0.y, s0.z) - O;
    float tca = dot(L, D);
    float d2 = dot(L, L) - tca * tca;
    if (d2 > s0.w) return false;
    float thc = sqrtf(s0.w - d2);
    float t0 = tca - thc, t1 = tca + thc;
    if (t0 > t1) {
        float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    if (t0 < 0.0f) {
        t0 = t1;
        if (t0 < 0.0f) return false;
    }
    t_out = t0;
    return true;
}

// compute.glsl:921-942
__device__ __forceinline__ bool box_distance(v3 O, v3 D, v3 b0, v3 b1, float &t_out) {
    v3 inv = V(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
    v3 tbot = inv * (b0 - O), ttop = inv * (b1 - O);
    v3 tmin = V(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    v3 tmax = V(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
    float t0 = fmaxf(fmaxf(tmin.x, tmin.y), fmaxf(tmin.x, tmin.z));
    float t1 = fminf(fminf(tmax.x, tmax.y), fminf(tmax.x, tmax.z));
    if (t1 <= fmaxf(t0, 0.0f)) return false;
    t_out = t0;
    return true;
}

#define CSWAP(ka, kb, ma, mb)          \
    {                                  \
        bool sw_ = (kb < ka);          \
        float tk_ = sw_ ? kb : ka;     \
        kb = sw_ ? ka : kb;            \
        ka = tk_;                      \
        uint32_t tm_ = sw_ ? mb : ma;  \
        mb = sw_ ? ma : mb;            \
        ma = tm_;                      \
    }

// One iteration of the reference's `while (stackPtr > 0)` loop: pop a node, slab-test its four
// child boxes, sort, visit far to near (leaves are intersected at once, inner children pushed).
// `stack` is this lane's column of the LDS stack (element stride STRIDE), `spill` its private
// overflow array.  best/best_code: distance and metadata word of the best hit so far.
// A child word of a device node record (type << 30 | position, see DevScene) as the reference's
// word (type << 30 | index); 0xFFFFFFFF, "no hit", stays.  A leaf index that lay beyond its array
// comes back as the index of the array's closing zero record.
__device__ __forceinline__ uint32_t reference_word(const DevScene &sc, uint32_t w) {
    const uint32_t type = w >> 30, unit = w & 0x3FFFFFFFu;
    const uint32_t idx = type == 0u ? (unit - sc.unit_tris) >> 1 : type == 1u ? unit - sc.unit_spheres
                       : type == 2u ? unit - sc.unit_boxes : unit >> 2;
    return w == 0xFFFFFFFFu ? w : (type << 30) | idx;
}

// The leaves of the node visited last that are still to be intersected, in one of two forms.
// As a list (the kernels with leaf steps of their own): v0.. are the pending leaves' words, v0 the
// next one, pend their number -- the next word is at hand when a leaf step starts, before its loads.
// As a mask (the unified step): v0..v3 are all four child words in visiting order (far to near) and
// bit j of pend says that v_j is a leaf still to be intersected -- building it costs a third of the
// instructions of the list, and the unified step picks its next item at the end of a step anyway.
struct LeafList {
    uint32_t v0, v1, v2, v3;
    uint32_t pend;
};

// Raw words of one node as loaded, before any arithmetic: the loads can be issued early and
// waited for only where node_test first touches them.
struct NodeRegs {
    float4 lox, hix, loy, hiy, loz, hiz, mt;
};

// The workgroup's LDS copy of the first nodes, typed as an LDS pointer: reads through it are LDS
// instructions (ds_read_b128), which bypass the vector-memory (texture address / data) path that
// bounds this kernel.  Through a generic pointer the compiler merges the LDS and the global case
// into flat loads, and every lane of those goes through the texture-address unit.
typedef float vec4f __attribute__((ext_vector_type(4)));  // plain vector: may carry an address space
using LdsTop = const __attribute__((address_space(3))) vec4f *;
using GlobalNodes = const __attribute__((address_space(1))) vec4f *;
using ConstVec = const __attribute__((address_space(4))) vec4f *;  // read-only for the kernel's lifetime: uniform addresses load through the scalar cache
__device__ __forceinline__ float4 ld4(LdsTop p) {
    vec4f v = *p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ld4(GlobalNodes p) {
    vec4f v = *p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// Issues the loads of node ni.  TOP > 0: the first TOP nodes (the top of the tree in the device
// numbering) are read from the workgroup's LDS copy `top` instead of through the vector
// memory pipeline, which is what bounds this kernel.
// FAST (the opt-in fast mode, not the reference's data): the node is one of the library's own
// 64-byte records (rt_fastbvh.cpp), 4 x 16 bytes: n0 = origin.xyz + exponents, n1 = lo_x hi_x lo_y
// hi_y codes, n2 = lo_z hi_z codes, n3 = child words; they travel in lox, hix, loy and mt.
template <int TOP = 0, bool FAST = false>
__device__ __forceinline__ void node_load(const DevScene &sc, uint32_t ni, NodeRegs &nd, LdsTop top = nullptr) {
    if (FAST) {
        ni = min(ni, sc.n_fnodes);  // out-of-range -> the all-empty sentinel node
        if (TOP > 0 && ni < (uint32_t)TOP) {
            LdsTop tp = top + ni * 4u;
            nd.lox = ld4(tp), nd.hix = ld4(tp + 1), nd.loy = ld4(tp + 2), nd.mt = ld4(tp + 3);
            return;
        }
        GlobalNodes np = (GlobalNodes)sc.fnodes + (size_t)ni * 4u;
        nd.lox = ld4(np), nd.hix = ld4(np + 1), nd.loy = ld4(np + 2), nd.mt = ld4(np + 3);
        return;
    }
    // ni: the node's position in units of 32 bytes (4 per node), never beyond the closing record
    if (TOP > 0) {
        // Written in assembly so that the two cases may be in flight together: in a wavefront whose
        // lanes split between them the compiler would wait for all global loads before it issues the
        // LDS reads, because both write the same registers (in different lanes, which it cannot know).
        // Between these statements and the waits below a0..a6 hold data still in flight, which the
        // compiler does not know either: tools/check_isa.py (make check-isa, part of the build)
        // verifies on the generated code that both arms got the same registers and that nothing it
        // placed in between (a copy at the merge, a spill) touches them.  (Issuing both kinds from ONE
        // statement under explicit execution masks, which needs no such check, costs 1.3 % of a
        // frame on cache-resident scenes.)
        vec4f a0, a1, a2, a3, a4, a5, a6;
        if (ni < 4u * (uint32_t)TOP) {
            const uint32_t addr = (uint32_t)(uintptr_t)top + ni * (RT_TOP_STRIDE * 4u);
            asm volatile(
                "ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:32\n\t"
                "ds_read_b128 %3, %7 offset:48\n\tds_read_b128 %4, %7 offset:64\n\tds_read_b128 %5, %7 offset:80\n\t"
                "ds_read_b128 %6, %7 offset:96"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6)
                : "v"(addr)
                : "memory");
        } else {
            const float4 *np = sc.nodes + (size_t)ni * 2u;
            asm volatile(
                "global_load_dwordx4 %0, %7, off\n\tglobal_load_dwordx4 %1, %7, off offset:16\n\t"
                "global_load_dwordx4 %2, %7, off offset:32\n\tglobal_load_dwordx4 %3, %7, off offset:48\n\t"
                "global_load_dwordx4 %4, %7, off offset:64\n\tglobal_load_dwordx4 %5, %7, off offset:80\n\t"
                "global_load_dwordx4 %6, %7, off offset:96"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6)
                : "v"(np)
                : "memory");
        }
        // either kind returns in order: vector k has arrived when at most 6 - k of its kind are outstanding
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(6)" : "+v"(a0));
        asm volatile("s_waitcnt vmcnt(5) lgkmcnt(5)" : "+v"(a1));
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(4)" : "+v"(a2));
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(3)" : "+v"(a3));
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(2)" : "+v"(a4));
        asm volatile("s_waitcnt vmcnt(1) lgkmcnt(1)" : "+v"(a5));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a6));
        nd.lox = make_float4(a0.x, a0.y, a0.z, a0.w), nd.hix = make_float4(a1.x, a1.y, a1.z, a1.w);
        nd.loy = make_float4(a2.x, a2.y, a2.z, a2.w), nd.hiy = make_float4(a3.x, a3.y, a3.z, a3.w);
        nd.loz = make_float4(a4.x, a4.y, a4.z, a4.w), nd.hiz = make_float4(a5.x, a5.y, a5.z, a5.w);
        nd.mt = make_float4(a6.x, a6.y, a6.z, a6.w);
        return;
    }
    GlobalNodes np = (GlobalNodes)sc.nodes + (size_t)ni * 2u;  // no LDS copy: plain loads
    nd.lox = ld4(np), nd.hix = ld4(np + 1), nd.loy = ld4(np + 2), nd.hiy = ld4(np + 3), nd.loz = ld4(np + 4),
    nd.hiz = ld4(np + 5), nd.mt = ld4(np + 6);
}

// Pops the word on top of the stack (entries are child words as the nodes hold them: 3 << 30 | node index).
template <int STRIDE, int LDS_DEPTH>
__device__ __forceinline__ uint32_t stack_pop(int &sp, const uint32_t *stack, const uint32_t *spill) {
    --sp;
    // the LDS row is read unconditionally (clamped), the private rows only by lanes that are
    // that deep: one plain LDS read on the common path instead of a generic-address load
    uint32_t ni = stack[min(sp, LDS_DEPTH - 1) * STRIDE];
    asm volatile("" : "+v"(ni));  // keeps the two reads from being merged into one generic load
    if (sp >= LDS_DEPTH) ni = spill[sp - LDS_DEPTH];
    return ni;
}

// Pops the next node index and issues its loads.
template <bool STATS, int STRIDE, int LDS_DEPTH, int TOP = 0, bool FAST = false>
__device__ __forceinline__ void node_fetch(const DevScene &sc, int &sp, const uint32_t *stack, const uint32_t *spill,
                                           NodeRegs &nd, LaneStats &st, LdsTop top = nullptr) {
    if (STATS) {
        st.max_sp = max(st.max_sp, (uint32_t)sp);
        st.nodes++;
    }
    node_load<TOP, FAST>(sc, stack_pop<STRIDE, LDS_DEPTH>(sp, stack, spill) & 0x3FFFFFFFu, nd, top);
}

// Slab tests, sort, pushes and leaf list for a fetched node.  FAST: the child boxes come from 8-bit
// codes, plane distance = fma(code, scale * ID, (origin - O) * ID) with ID kept finite by the
// caller, the exit distance is widened by 4 ulp, and an empty child slot (0xFFFFFFFF) is never valid.
template <bool STATS, int STRIDE, int LDS_DEPTH, bool FAST = false, bool MASK = false>
__device__ __forceinline__ void node_test(const DevScene &sc, const NodeRegs &nd, v3 O, v3 ID, float best, int &sp,
                                          uint32_t *stack, uint32_t *spill, LeafList &ll, LaneStats &st) {
    uint32_t m0 = f2u(nd.mt.x), m1 = f2u(nd.mt.y), m2 = f2u(nd.mt.z), m3 = f2u(nd.mt.w);
    // A child that is not visited has its metadata word replaced by kSkip before the sort: the sort
    // compares distances only, so the words travel exactly as before, with one word per child to
    // move instead of two.  No child word equals kSkip by itself: the repack maps out-of-range node
    // indices (the reference's empty slot, 0xFFFFFFFF) to the sentinel node, whose index is lower.
    constexpr uint32_t kSkip = 0xFFFFFFFFu;
    float k0, k1, k2, k3;
    if (FAST) {
        const uint32_t ew = f2u(nd.lox.w);
        const float ax = u2f((ew & 0xFFu) << 23) * ID.x, ay = u2f(((ew >> 8) & 0xFFu) << 23) * ID.y,
                    az = u2f(((ew >> 16) & 0xFFu) << 23) * ID.z;
        const float bx = (nd.lox.x - O.x) * ID.x, by = (nd.lox.y - O.y) * ID.y, bz = (nd.lox.z - O.z) * ID.z;
        const uint32_t qlx = f2u(nd.hix.x), qhx = f2u(nd.hix.y), qly = f2u(nd.hix.z), qhy = f2u(nd.hix.w),
                       qlz = f2u(nd.loy.x), qhz = f2u(nd.loy.y);
        // The plane a ray enters a slab through is the low one where the direction is positive and the
        // high one where it is negative: lo code <= hi code and the scale is positive, so
        // fma(code, a, b) is monotone in the code and min(t_lo, t_hi) / max(t_lo, t_hi) are simply the
        // one or the other, bit for bit (ID is finite and non-zero here, so nothing is NaN).  Choosing
        // the code words once per node saves the six min / max per child.
        const bool px = ID.x >= 0.0f, py = ID.y >= 0.0f, pz = ID.z >= 0.0f;
        const uint32_t nx = px ? qlx : qhx, fx = px ? qhx : qlx, ny = py ? qly : qhy, fy = py ? qhy : qly,
                       nz = pz ? qlz : qhz, fz = pz ? qhz : qlz;
#define FSLAB(SH, M, KEY)                                                                                  \
    {                                                                                                       \
        float tnx = __builtin_fmaf((float)((nx >> SH) & 0xFFu), ax, bx), tfx = __builtin_fmaf((float)((fx >> SH) & 0xFFu), ax, bx); \
        float tny = __builtin_fmaf((float)((ny >> SH) & 0xFFu), ay, by), tfy = __builtin_fmaf((float)((fy >> SH) & 0xFFu), ay, by); \
        float tnz = __builtin_fmaf((float)((nz >> SH) & 0xFFu), az, bz), tfz = __builtin_fmaf((float)((fz >> SH) & 0xFFu), az, bz); \
        float tN = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));                                                \
        float tF = fminf(fminf(tfx, tfy), tfz) * 1.0000004f;                                                \
        KEY = tN;                                                                                           \
        M = ((tN <= tF) && (tN < best)) ? M : kSkip; /* an empty slot is kSkip already */                   \
    }
        FSLAB(0, m0, k0) FSLAB(8, m1, k1) FSLAB(16, m2, k2) FSLAB(24, m3, k3)
#undef FSLAB
    } else {
    float4 lox = nd.lox, hix = nd.hix, loy = nd.loy, hiy = nd.hiy, loz = nd.loz, hiz = nd.hiz;
    // compute.glsl:186-208, one child per macro expansion
#define SLAB(c, KEY, M)                                                                          \
    {                                                                                            \
        float t1x = (lox.c - O.x) * ID.x, t2x = (hix.c - O.x) * ID.x;                            \
        float t1y = (loy.c - O.y) * ID.y, t2y = (hiy.c - O.y) * ID.y;                            \
        float t1z = (loz.c - O.z) * ID.z, t2z = (hiz.c - O.z) * ID.z;                            \
        float tN = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fmaxf(fminf(t1z, t2z), 0.0f)); \
        float tF = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));              \
        KEY = tN;                                                                                \
        M = ((tN <= tF) && (tN < best)) ? M : kSkip;                                             \
    }
    SLAB(x, k0, m0) SLAB(y, k1, m1) SLAB(z, k2, m2) SLAB(w, k3, m3)
#undef SLAB
    }
    // compute.glsl:210-223
    CSWAP(k0, k1, m0, m1)
    CSWAP(k2, k3, m2, m3)
    CSWAP(k0, k2, m0, m2)
    CSWAP(k1, k3, m1, m3)
    CSWAP(k1, k2, m1, m2)
    // compute.glsl:290-326: children are visited far to near; an inner child is pushed, a leaf is
    // intersected at once.  Pushing and intersecting do not influence each other, so the visit is
    // split in two passes that keep each kind in the reference's order: first the pushes
    // (k = 3..0), then the leaves (k = 3..0) from a per-lane list.  The wavefront then runs the
    // intersection code once per list position instead of once per child slot.
    // inner and visited: a node word other than kSkip
    const uint32_t in3 = (m3 - 0xC0000000u < 0x3FFFFFFFu) ? 1u : 0u, in2 = (m2 - 0xC0000000u < 0x3FFFFFFFu) ? 1u : 0u;
    const uint32_t in1 = (m1 - 0xC0000000u < 0x3FFFFFFFu) ? 1u : 0u, in0 = (m0 - 0xC0000000u < 0x3FFFFFFFu) ? 1u : 0u;
    {
        // Rows the four children would take, in visiting order k = 3..0: every child is stored at the
        // running top and only an inner, visited child advances it, so a skipped child is overwritten
        // by the next one or left above the final top -- no branch on the child's kind.  A row at or
        // beyond LDS_DEPTH is not in LDS: those stores are masked out here and redone into private
        // memory below, which few lanes ever need (one pop in a hundred comes from that deep).
        const int r3 = sp, r2 = r3 + (int)in3, r1 = r2 + (int)in2, r0 = r1 + (int)in1, top_after = r0 + (int)in0;
        if (r3 < LDS_DEPTH) stack[r3 * STRIDE] = m3;
        if (r2 < LDS_DEPTH) stack[r2 * STRIDE] = m2;
        if (r1 < LDS_DEPTH) stack[r1 * STRIDE] = m1;
        if (r0 < LDS_DEPTH) stack[r0 * STRIDE] = m0;
        sp = top_after;
        if (top_after > LDS_DEPTH) {
            // a push onto a full stack is dropped and counted, and does not advance the top
#define PUSH_DEEP(IN, META, ROW)                                                                  \
    if (IN && ROW >= LDS_DEPTH) {                                                                 \
        if (ROW < LDS_DEPTH + RT_STACK_SPILL) spill[ROW - LDS_DEPTH] = META;                      \
        else st.overflow++;                                                                       \
    }
            PUSH_DEEP(in3, m3, r3) PUSH_DEEP(in2, m2, r2) PUSH_DEEP(in1, m1, r1) PUSH_DEEP(in0, m0, r0)
#undef PUSH_DEEP
            sp = min(top_after, LDS_DEPTH + RT_STACK_SPILL);
        }
    }
    // the leaves among the visited children (a word below 0xC0000000), k = 3 first
    if (MASK) {
        // the words stay where the sort left them, one bit per word marks the leaves
        ll.v0 = m3, ll.v1 = m2, ll.v2 = m1, ll.v3 = m0;
        // (doubling and adding the next bit is one add-with-carry per word)
        uint32_t pend = m0 < 0xC0000000u ? 1u : 0u;
        pend = pend + pend + (m1 < 0xC0000000u ? 1u : 0u);
        pend = pend + pend + (m2 < 0xC0000000u ? 1u : 0u);
        pend = pend + pend + (m3 < 0xC0000000u ? 1u : 0u);
        ll.pend = pend;
        return;
    }
    // build the list by prepending in the order k = 0..3, so that k = 3 ends up in front
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0, nl = 0;
#define ADD_LEAF(META)                                                                            \
    {                                                                                             \
        bool lf = META < 0xC0000000u;                                                             \
        nl += lf ? 1u : 0u;                                                                       \
        l3 = lf ? l2 : l3;                                                                        \
        l2 = lf ? l1 : l2;                                                                        \
        l1 = lf ? l0 : l1;                                                                        \
        l0 = lf ? META : l0;                                                                      \
    }
    ADD_LEAF(m0) ADD_LEAF(m1) ADD_LEAF(m2) ADD_LEAF(m3)
#undef ADD_LEAF
    ll.v0 = l0, ll.v1 = l1, ll.v2 = l2, ll.v3 = l3;
    ll.pend = nl;
}

template <bool STATS, int STRIDE, int LDS_DEPTH, int TOP = 0, bool FAST = false>
__device__ __forceinline__ void node_part(const DevScene &sc, v3 O, v3 ID, float best, int &sp, uint32_t *stack,
                                          uint32_t *spill, LeafList &ll, LaneStats &st, LdsTop top = nullptr) {
    NodeRegs nd;
    node_fetch<STATS, STRIDE, LDS_DEPTH, TOP, FAST>(sc, sp, stack, spill, nd, st, top);
    node_test<STATS, STRIDE, LDS_DEPTH, FAST>(sc, nd, O, ID, best, sp, stack, spill, ll, st);
}

// Intersects the first pending leaf of the list (compute.glsl:299-322).
template <bool STATS, bool FAST = false>
__device__ __forceinline__ void leaf_one(const DevScene &sc, v3 O, v3 D, float &best, uint32_t &best_code,
                                         LeafList &ll, LaneStats &st) {
    uint32_t meta = ll.v0;  // list form
    ll.v0 = ll.v1;
    ll.v1 = ll.v2;
    ll.v2 = ll.v3;
    ll.pend--;
    uint32_t type = meta >> 30, idx = meta & 0x3FFFFFFFu;
    float t;
    bool hit;
    // the words of the library's own tree (FAST) hold indices; those of the device node records positions
    const float4 *rec = sc.nodes + (size_t)idx * 2u;
    if (type == 0u) {
        const float4 *tp = FAST ? sc.tris + (size_t)min(idx, sc.n_tris) * kTriVec : rec;
        if (STATS) st.tri++;
        hit = tri_distance(O, D, tp[0], tp[1], tp[2], t);
    } else if (type == 1u) {
        const float4 *spn = FAST ? sc.spheres + (size_t)min(idx, sc.n_spheres) * kSphereVec : rec;
        if (STATS) st.sph++;
        hit = sphere_distance(O, D, spn[0], t);
    } else {
        const float4 *bp = FAST ? sc.boxes + (size_t)min(idx, sc.n_boxes) * 2 : rec;
        float4 b0 = bp[0], b1 = bp[1];
        if (STATS) st.box++;
        hit = box_distance(O, D, V(b0.x, b0.y, b0.z), V(b1.x, b1.y, b1.z), t);
    }
    if (hit && t > 0.0f && t < best - 1e-6f) {
        best = t;
        best_code = meta;
    }
}

// One iteration of the reference's `while (stackPtr > 0)` loop.
template <bool STATS, int STRIDE, int LDS_DEPTH, int TOP = 0, bool FAST = false>
__device__ __forceinline__ void node_step(const DevScene &sc, v3 O, v3 D, v3 ID, float &best, uint32_t &best_code,
                                          int &sp, uint32_t *stack, uint32_t *spill, LaneStats &st,
                                          LdsTop top = nullptr) {
    LeafList ll;
    node_part<STATS, STRIDE, LDS_DEPTH, TOP, FAST>(sc, O, ID, best, sp, stack, spill, ll, st, top);
    while (ll.pend != 0u) leaf_one<STATS, FAST>(sc, O, D, best, best_code, ll, st);
}

// The whole traversal of one ray (compute.glsl:225-330) with ONE kind of step for nodes and
// leaves: every lane holds the item it works on next -- a node to visit or a leaf primitive to
// intersect -- and a step fetches the item's record and processes it.  A lane's own sequence (node,
// its leaves far to near, next node, ...) is exactly that of node_step, so the results are the
// same; what changes is what the wavefront pays.  The CU's vector-memory return path charges 16
// cycles per load instruction however few lanes are active and waits for the cache once per batch
// of loads that miss; with leaves intersected in steps of their own, a wavefront issued 7 loads per
// node step plus 3 per leaf step, the latter for 6 active lanes on average.  Here the lanes that
// are at a leaf ride along in the 7 loads of the lanes that are at a node: the loads fetch the
// first 112 bytes of the 128-byte line the item's record lies in (records never straddle lines:
// nodes take a line, two triangles or four spheres share one), and a leaf lane picks its record
// out of them.  Boxes (two records of four would not fit) keep loads of their own.
// Where a lane stands in the traversal of its ray: the item it works on next (a child word: a node
// to visit or a leaf primitive to intersect), the pending leaves of the node visited last, the stack
// height.  `have` is false once the ray is done.
struct Walk {
    uint32_t cur;
    LeafList ll;
    int sp;
    bool have;
};
template <bool STATS>
__device__ __forceinline__ void walk_begin(Walk &w, LaneStats &st) {
    w.cur = 0xC0000000u;  // the root node
    w.ll = {0u, 0u, 0u, 0u, 0u};
    w.sp = 0;
    w.have = true;
    if (STATS) st.max_sp = max(st.max_sp, 1u);  // the reference pushes the root before its loop
}

// One step of traverse_unified (below) for the lanes that call it.
template <bool STATS, int STRIDE, int LDS_DEPTH, int TOP>
__device__ __forceinline__ void walk_step(const DevScene &sc, v3 O, v3 D, v3 ID, float &best, uint32_t &best_code,
                                          Walk &w, uint32_t *stack, uint32_t *spill, LaneStats &st, LdsTop top) {
    uint32_t &cur = w.cur;
    LeafList &ll = w.ll;
    int &sp = w.sp;
    bool &have = w.have;
    {
        const uint32_t type = cur >> 30, idx = cur & 0x3FFFFFFFu;
        // idx: the record's position in units of 32 bytes from the first node (4 per node)
        const bool in_lds = type == 3u && idx < 4u * (uint32_t)TOP;
        vec4f a0, a1, a2, a3, a4, a5, a6;  // in flight until the waits below: see node_load and tools/check_isa.py
        uintptr_t rec;
        // The slowest ray of 64 sets the pace of a bounce: a wavefront spends a tenth of its steps with a
        // single ray left.  That ray's record then comes through the scalar data cache (read only; the
        // address made uniform with readfirstlane): such a step costs the vector-memory return path,
        // which bounds this kernel, nothing.  The values stay in scalar registers until the waits
        // below are over, so nothing of this arm touches a vector register that the other arms load
        // into.  (RT_SOLO_SCALAR=0: A/B builds without it.  Measured and not kept: the scalar path
        // for two to four lanes one after the other, and for one lane off the LDS copy while others
        // are on it -- both slower than the vector loads.)
#ifndef RT_SOLO_SCALAR
#define RT_SOLO_SCALAR 1
#endif
#ifndef RT_MASK_REST
#define RT_MASK_REST 1
#endif
        const uint64_t live = __builtin_amdgcn_ballot_w64(true);
        const bool solo = RT_SOLO_SCALAR && (live & (live - 1u)) == 0u;
        // the address of the item's record (whatever its kind) and of the two runs of 16-byte vectors
        // fetched for it.  Seven loads cover the first 112 bytes of the record's 128-byte line, except
        // for a triangle: its first three start at the record itself, so that its 48 bytes arrive in
        // a0..a2 whichever half of the line it lies in, and the other four, which its lane does not
        // use, re-read the start of the line -- no second line is touched, nothing is read beyond
        // a buffer.  (Only the low address word differs: records do not straddle lines.)
        rec = (uintptr_t)(sc.nodes + (size_t)idx * 2u);
        const uint32_t rec_lo = (uint32_t)rec, line_lo = rec_lo & ~127u;
        const uint32_t first_lo = type == 0u ? rec_lo : line_lo, rest_lo = type == 0u ? line_lo : line_lo + 48u;
        vec4f c0, c1, c2, c3, c4, c5, c6;  // the scalar arm's copy of the record
        bool from_scalar = false;
        if (in_lds) {
            const uint32_t addr = (uint32_t)(uintptr_t)top + idx * (RT_TOP_STRIDE * 4u);
            asm volatile(
                "ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:32\n\t"
                "ds_read_b128 %3, %7 offset:48\n\tds_read_b128 %4, %7 offset:64\n\tds_read_b128 %5, %7 offset:80\n\t"
                "ds_read_b128 %6, %7 offset:96"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6)
                : "v"(addr)
                : "memory");
        } else if (solo) {
            const uintptr_t base = (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(rec >> 32)) << 32;
            ConstVec first = (ConstVec)(base | (uint32_t)__builtin_amdgcn_readfirstlane(first_lo));
            ConstVec rest = (ConstVec)(base | (uint32_t)__builtin_amdgcn_readfirstlane(rest_lo));
            c0 = first[0], c1 = first[1], c2 = first[2], c3 = rest[0], c4 = rest[1], c5 = rest[2], c6 = rest[3];
            from_scalar = true;
        } else {
            const uintptr_t hi = rec & ~(uintptr_t)0xFFFFFFFFu;
            const uintptr_t first = hi | first_lo, rest = hi | rest_lo;
#if RT_MASK_REST
            // The last four loads run only in the lanes whose item is not a triangle (a triangle's
            // lane never looks at a3..a6): every lane of a load instruction is one more access for the
            // CU's L1 to look up, whatever it returns.
            const uint64_t wide = __builtin_amdgcn_ballot_w64(type != 0u);
            uint64_t all_lanes;
            asm volatile(
                "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:16\n\t"
                "global_load_dwordx4 %2, %8, off offset:32\n\t"
                "s_mov_b64 %7, exec\n\ts_and_b64 exec, exec, %10\n\t"
                "global_load_dwordx4 %3, %9, off\n\t"
                "global_load_dwordx4 %4, %9, off offset:16\n\tglobal_load_dwordx4 %5, %9, off offset:32\n\t"
                "global_load_dwordx4 %6, %9, off offset:48\n\t"
                "s_mov_b64 exec, %7"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&s"(all_lanes)
                : "v"(first), "v"(rest), "s"(wide)
                : "memory", "scc");
#else
            asm volatile(
                "global_load_dwordx4 %0, %7, off\n\tglobal_load_dwordx4 %1, %7, off offset:16\n\t"
                "global_load_dwordx4 %2, %7, off offset:32\n\tglobal_load_dwordx4 %3, %8, off\n\t"
                "global_load_dwordx4 %4, %8, off offset:16\n\tglobal_load_dwordx4 %5, %8, off offset:32\n\t"
                "global_load_dwordx4 %6, %8, off offset:48"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6)
                : "v"(first), "v"(rest)
                : "memory");
#endif
        }
        // either kind returns in order: vector k has arrived when at most 6 - k of its kind are outstanding
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(6)" : "+v"(a0));
        asm volatile("s_waitcnt vmcnt(5) lgkmcnt(5)" : "+v"(a1));
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(4)" : "+v"(a2));
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(3)" : "+v"(a3));
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(2)" : "+v"(a4));
        asm volatile("s_waitcnt vmcnt(1) lgkmcnt(1)" : "+v"(a5));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a6));
        if (from_scalar) a0 = c0, a1 = c1, a2 = c2, a3 = c3, a4 = c4, a5 = c5, a6 = c6;
#define F4(a) make_float4(a.x, a.y, a.z, a.w)
        if (type == 3u) {
            if (STATS) st.nodes++;
            NodeRegs nd;
            nd.lox = F4(a0), nd.hix = F4(a1), nd.loy = F4(a2), nd.hiy = F4(a3), nd.loz = F4(a4), nd.hiz = F4(a5), nd.mt = F4(a6);
            node_test<STATS, STRIDE, LDS_DEPTH, false, true>(sc, nd, O, ID, best, sp, stack, spill, ll, st);
        } else {
            float t;
            bool hit;
            if (type == 0u) {
                if (STATS) st.tri++;
                hit = tri_distance(O, D, F4(a0), F4(a1), F4(a2), t);
            } else if (type == 1u) {
                const uint32_t q = (uint32_t)(rec >> 5) & 3u;  // which of the four records of its line
                const vec4f s0 = q == 0u ? a0 : q == 1u ? a2 : q == 2u ? a4 : a6;
                if (STATS) st.sph++;
                hit = sphere_distance(O, D, F4(s0), t);
            } else {
                const float4 *bp = reinterpret_cast<const float4 *>(rec);
                float4 b0 = bp[0], b1 = bp[1];
                if (STATS) st.box++;
                hit = box_distance(O, D, V(b0.x, b0.y, b0.z), V(b1.x, b1.y, b1.z), t);
            }
            if (hit && t > 0.0f && t < best - 1e-6f) {
                best = t;
                best_code = cur;
            }
        }
#undef F4
        // the next item: the nearest pending leaf of the node just visited, else the next node
        // (written with selects and one converging branch: the loop must keep a single back edge, or
        // the compiler nests it and the lanes at a leaf hold up the lanes at a node again)
        const bool from_list = ll.pend != 0u;
        const bool from_stack = !from_list && sp > 0;
        uint32_t popped = 0u;
        if (from_stack) {
            if (STATS) st.max_sp = max(st.max_sp, (uint32_t)sp);
            popped = stack_pop<STRIDE, LDS_DEPTH>(sp, stack, spill);
        }
        // (with pend == 0 nothing is taken from the list: whatever the selection yields is dropped)
        const int j = __builtin_ctz(ll.pend);
        const uint32_t leaf_a = j == 0 ? ll.v0 : ll.v1, leaf_b = j == 2 ? ll.v2 : ll.v3;
        const uint32_t leaf = j < 2 ? leaf_a : leaf_b;
        ll.pend &= ll.pend - 1u;
        cur = from_list ? leaf : popped;
        have = from_list || from_stack;
    }
}

template <bool STATS, int STRIDE, int LDS_DEPTH, int TOP>
__device__ __forceinline__ void traverse_unified(const DevScene &sc, v3 O, v3 D, v3 ID, float &best,
                                                 uint32_t &best_code, uint32_t *stack, uint32_t *spill, LaneStats &st,
                                                 LdsTop top) {
    Walk w;
    walk_begin<STATS>(w, st);
    // Nothing of this lane's earlier vector-memory traffic may still be pending, in the compiler's
    // books, when the loop is entered: it cannot see the hand-written loads of a step, so a load it
    // believes outstanding from before the loop (a register it must not overwrite yet) makes it put
    // a vmcnt(0) in front of the LDS arm's reads -- where, in a wavefront whose lanes split between
    // the arms, it waits for the other lanes' global loads issued a moment earlier.  An explicit
    // wait here (encoding: vmcnt(0), the other counters at their maxima) keeps those out of the loop.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    while (w.have) walk_step<STATS, STRIDE, LDS_DEPTH, TOP>(sc, O, D, ID, best, best_code, w, stack, spill, st, top);
    best_code = reference_word(sc, best_code);
}

// Surface attributes of the accepted primitive, derived from (O, D, t, code) exactly as the
// intersection routines derive them (compute.glsl:838-843, :909-916, :944-961).
__device__ __forceinline__ void hit_attributes(const DevScene &sc, v3 O, v3 D, float t, uint32_t code, v3 &pos,
                                               v3 &n, uint32_t &mat) {
    uint32_t type = code >> 30, idx = code & 0x3FFFFFFFu;
    pos = O + D * t;
    if (type == 0u) {
        const float4 *tp = sc.tris + (size_t)min(idx, sc.n_tris) * kTriVec;
        float4 r3 = tp[3];
        n = V(r3.x, r3.y, r3.z);
        mat = f2u(r3.w);
        if (dot(D, n) > 0.0f) n = -n;
    } else if (type == 1u) {
        const float4 *spn = sc.spheres + (size_t)min(idx, sc.n_spheres) * kSphereVec;
        float4 s0 = spn[0], s1 = spn[1];
        n = normalize(pos - V(s0.x, s0.y, s0.z));
        mat = f2u(s1.x);
        if (dot(D, n) > 0.0f) n = -n;
    } else {
        const float4 *bp = sc.boxes + (size_t)min(idx, sc.n_boxes) * 2;
        float4 b0 = bp[0], b1 = bp[1];
        mat = f2u(b0.w);
        const float e = 0.00001f;
        n = V(0.f, 0.f, 0.f);  // the reference leaves it undefined when no face is within epsilon
        if (fabsf(pos.x - b0.x) < e) n = V(-1.f, 0.f, 0.f);
        else if (fabsf(pos.x - b1.x) < e) n = V(1.f, 0.f, 0.f);
        else if (fabsf(pos.y - b0.y) < e) n = V(0.f, -1.f, 0.f);
        else if (fabsf(pos.y - b1.y) < e) n = V(0.f, 1.f, 0.f);
        else if (fabsf(pos.z - b0.z) < e) n = V(0.f, 0.f, -1.f);
        else if (fabsf(pos.z - b1.z) < e) n = V(0.f, 0.f, 1.f);
    }
}

// compute.glsl:767-805.  The reference's loop bound is inclusive; the extra record reads as zeros
// (robust buffer access), i.e. a plane with a zero normal: denom is +-0, the test -denom > 1e-7
// fails, and the iteration changes nothing.  It is therefore left out.
__device__ __forceinline__ void planes_nearest(const DevScene &sc, v3 O, v3 D, bool &exists, float &t_best, v3 &pos,
                                               v3 &n, uint32_t &mat) {
    exists = false;
    t_best = 9999999.0f;
    for (uint32_t i = 0; i < sc.n_planes; i++) {
        float4 p0 = sc.planes[(size_t)i * 2], p1 = sc.planes[(size_t)i * 2 + 1];
        v3 pp = V(p0.x, p0.y, p0.z), pn = V(p1.x, p1.y, p1.z);
        float denom = dot(pn, D);
        if (-denom > 0.0000001f) {
            float t = dot(pp - O, pn) / denom;
            if (t > 0.0f && t < t_best) {
                exists = true;
                t_best = t;
                n = pn;
                pos = O + D * t;
                mat = f2u(p0.w);
            }
        }
    }
}

// compute.glsl:392-406, :381-388, :577.  sin_theta is sqrt(1 - x) with x the argument of
// cos_theta's square root (sqrt(x)*sqrt(x) folds to x), and 1 + (a2-1)*g2 is (1-g2) + a2*g2.
__device__ __forceinline__ v3 ggx_half_vector(v3 n, float a2, float g1, float g2) {
    float phi = (2.0f * 3.14159265f) * g1;
    float omg = 1.0f - g2;
    float ct2 = omg / (omg + a2 * g2);
    float ct = sqrtf(ct2);
    float st = sqrtf(1.0f - fabsf(ct2));
    float sn, cs;
    d_sincos(phi, sn, cs);
    float hx = cs * st, hy = sn * st, hz = ct;
    v3 T;
    if (fabsf(n.x) > fabsf(n.y)) T = V(-n.z, 0.0f, n.x) / sqrtf(n.x * n.x + n.z * n.z);
    else T = V(0.0f, n.z, -n.y) / sqrtf(n.y * n.y + n.z * n.z);
    v3 B = cross(n, T);
    return normalize((T * hx + B * hy) + n * hz);
}

// ------------------------------------------------------------------------------------------
// per-path state and the two halves of main() that do not traverse: ray generation
// (compute.glsl:459-494) and the bounce body after the hit is known (:500-664)
// ------------------------------------------------------------------------------------------
struct Path {
    v3 O, D;
    v3 thr, light;
    uint32_t rng;
    float ior;
};

__device__ __forceinline__ void primary_ray(const DevScene &sc, int W, int H, int x, int y, float time, Path &ps) {
    float px = (float)(2 * x - W) / (float)W;
    float py = (float)(2 * y - H) / (float)H;
    // uint(time * 1000): beyond the range of uint (frame 4 294 968 on) the reference shader's
    // conversion saturates, on llvmpipe as on GPUs; written out rather than left to the cast
    const float ms = time * 1000.0f;
    const uint32_t tw = ms >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)ms;
    uint32_t state = ((uint32_t)x * 1973u + (uint32_t)y * 9277u + tw * 26699u) | 1u;
    float jx = draw(state) - 0.5f;
    float jy = draw(state) - 0.5f;
    float qx = px + jx / (float)W;
    float qy = py + jy / (float)H;
    const float *M = sc.inv_vp;
    float far4[4], near4[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        far4[r] = ((M[r] * qx + M[4 + r] * qy) + M[8 + r] * 1.0f) + M[12 + r] * 1.0f;
        near4[r] = ((M[r] * qx + M[4 + r] * qy) + M[8 + r] * 0.0f) + M[12 + r] * 1.0f;
    }
    v3 farp = V(far4[0] / far4[3], far4[1] / far4[3], far4[2] / far4[3]);
    v3 nearp = V(near4[0] / near4[3], near4[1] / near4[3], near4[2] / near4[3]);
    ps.D = normalize(farp - nearp);
    ps.O = V(sc.eye[0], sc.eye[1], sc.eye[2]);
    ps.thr = V(1.f, 1.f, 1.f);
    ps.light = V(0.f, 0.f, 0.f);
    ps.rng = state;
    ps.ior = 1.0f;
}

// The body of one loop iteration after `hit = calculate_ray_bvh(ray)`.  tb/code: result of the
// BVH traversal.  Returns false when the path ends here (miss, or Russian roulette).  `exists`
// and `thit` report the hit the iteration saw (for the parity record).
template <bool STATS>
__device__ __forceinline__ bool bounce(const DevScene &sc, Path &ps, float tb, uint32_t code, uint32_t i,
                                       bool &exists, float &thit, LaneStats &st) {
    v3 O = ps.O, D = ps.D;
    // compute.glsl:332-351: the BVH hit wins only if nearer than the nearest plane hit
    bool p_exists;
    float tp;
    v3 pos, n;
    uint32_t mat;
    planes_nearest(sc, O, D, p_exists, tp, pos, n, mat);
    if (tb < tp) {
        exists = true;  // best < 9999999 implies a primitive was accepted
        thit = tb;
        hit_attributes(sc, O, D, tb, code, pos, n, mat);
    } else {
        exists = p_exists;
        thit = tp;
    }
    if (!exists) return false;
    if (STATS) st.hits++;
    const float4 *mp = sc.mats + (size_t)min(mat, sc.n_mats) * kMatVec;
    float4 q0 = mp[0], q1 = mp[1], q2 = mp[2], q3 = mp[3];
    float smooth = q0.w, trans = q1.w, f0 = q2.w, ior = q3.x, a2 = q3.y;
    ps.light = ps.light + V(q0.x, q0.y, q0.z) * ps.thr;
    float cosT = fminf(dot(-D, n), 1.0f);
    float R = f0 + (1.0f - f0) * d_pow(1.0f - cosT, 5.0f);
    v3 Dn;
    if (draw(ps.rng) < R) {
        if (smooth == 1.0f) {
            Dn = reflect(D, n);
        } else {
            float g1 = draw(ps.rng), g2 = draw(ps.rng);
            v3 Hh = ggx_half_vector(n, a2, g1, g2);
            Dn = reflect(D, Hh);
        }
        ps.O = pos + Dn * 0.0001f;
        ps.thr = ps.thr * V(q1.x, q1.y, q1.z);
    } else if (draw(ps.rng) < trans) {
        bool entering = dot(D, n) < 0.0f;
        v3 nn = entering ? n : -n;
        float n1 = ps.ior, n2 = entering ? ior : 1.0f;
        float g1 = draw(ps.rng), g2 = draw(ps.rng);
        v3 Hh = ggx_half_vector(n, a2, g1, g2);
        float eta = n1 / n2;
        float dn = dot(Hh, D);
        float k = 1.0f - eta * (eta * (1.0f - dn * dn));  // association of the GLSL built-in
        v3 r = V(0.f, 0.f, 0.f);
        if (!(k < 0.0f)) r = D * eta - Hh * (eta * dn + sqrtf(k));
        if (dot(r, r) == 0.0f) {
            Dn = reflect(D, nn);
        } else {
            Dn = r;
            ps.ior = n2;
        }
        ps.O = pos + Dn * 0.0001f;
    } else {
        float u1 = draw(ps.rng), u2 = draw(ps.rng);
        float rr = sqrtf(u2);
        float th = (2.0f * 3.14159265f) * u1;
        float sn, cs;
        d_sincos(th, sn, cs);
        float sx = rr * cs, sy = rr * sn, sz = sqrtf(fmaxf(0.0f, 1.0f - u2));
        v3 tg;
        if (fabsf(n.x) > fabsf(n.y)) {
            v3 t0 = V(n.z, 0.0f, -n.x);
            tg = t0 / sqrtf(dot(t0, t0));
        } else {
            v3 t0 = V(0.0f, -n.z, n.y);
            tg = t0 / sqrtf(dot(t0, t0));
        }
        v3 bt = cross(n, tg);
        v3 dif = (tg * sx + bt * sy) + n * sz;
        float dDn = dot(D, n);
        v3 spec = normalize(D - (n * 2.0f) * dDn);
        ps.O = pos + n * 0.0001f;
        Dn = V(dif.x + smooth * (spec.x - dif.x), dif.y + smooth * (spec.y - dif.y),
               dif.z + smooth * (spec.z - dif.z));
        if (dot(Dn, n) < 0.0f) Dn = -Dn;
        ps.thr = ps.thr * V(q2.x, q2.y, q2.z);
    }
    ps.D = Dn;
    // compute.glsl:428-445
    if (i >= 3u) {
        if (i >= 64u) return false;
        float lum = dot(ps.thr, V(0.2126f, 0.7152f, 0.0722f));
        float pr = fminf(fmaxf(lum, 0.05f), 0.95f);
        if (draw(ps.rng) > pr) return false;
        ps.thr = ps.thr / pr;
    }
    return true;
}

__device__ __forceinline__ uint32_t morton2(uint32_t x, uint32_t y) {
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

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor(v, o));
    return v;
}

// one atomic per wavefront per counter, on the stripe of this workgroup
template <bool STATS>
__device__ __forceinline__ void flush_counters(Counters *base, uint32_t lane, uint32_t nrays, const LaneStats &st) {
    Counters *c = base + (blockIdx.x & (uint32_t)(kCounterStripes - 1));
    uint32_t r = wave_sum(nrays), ov = wave_sum(st.overflow);
    if (lane == 0) {
        if (r) atomicAdd(&c->rays, (unsigned long long)r);
        if (ov) atomicAdd(&c->stack_overflows, (unsigned long long)ov);
    }
    if (STATS) {
        uint32_t a = wave_sum(st.nodes), t3 = wave_sum(st.tri), s3 = wave_sum(st.sph), b3 = wave_sum(st.box),
                 m = wave_max(st.max_sp), h3 = wave_sum(st.hits);
        if (lane == 0) {
            atomicAdd(&c->nodes, (unsigned long long)a);
            atomicAdd(&c->tri_tests, (unsigned long long)t3);
            atomicAdd(&c->sphere_tests, (unsigned long long)s3);
            atomicAdd(&c->box_tests, (unsigned long long)b3);
            atomicMax(&c->max_stack, (unsigned long long)m);
            atomicAdd(&c->hits, (unsigned long long)h3);
        }
    }
}

// tile -> pixel mapping shared by the megakernel and the ray generator: 16x16 pixel tile per
// workgroup, 8x8 per wavefront.  Workgroups that share an XCD (equal blockIdx % 8 under
// round-robin placement) get a contiguous run of tiles so that their node working sets overlap
// in that XCD's L2.  Returns false for surplus workgroups and uncovered / unowned pixels.
template <int BLOCK = RT_BLOCK>
__device__ __forceinline__ bool tile_pixel(const TraceParams &p, bool sharded, int &x, int &y) {
    // BLOCK threads = BLOCK/64 wavefronts of a tile; the workgroups of one tile are consecutive
    // within their XCD's run
    constexpr uint32_t WPB = BLOCK / 64, GROUPS = 4 / WPB;  // wavefronts per workgroup, workgroups per tile
    uint32_t b = blockIdx.x;
    uint32_t per = (p.ntiles + 7u) / 8u;
    uint32_t j = b >> 3;
    uint32_t tile = (b & 7u) * per + j / GROUPS;
    if (tile >= p.ntiles) return false;
    uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    uint32_t lane = threadIdx.x & 63u, wave = (j % GROUPS) * WPB + (threadIdx.x >> 6);
    x = (int)(tx * 16u + (wave & 1u) * 8u + (lane & 7u));
    y = (int)(ty * 16u + (wave >> 1) * 8u + (lane >> 3));
    bool live = (x < p.WW) && (y < p.HH);
    if (sharded && p.nshards > 1u) {
        // 64x64 pixel macro tiles dealt round-robin in Morton order
        uint32_t owner = morton2((uint32_t)x >> 6, (uint32_t)y >> 6) % p.nshards;
        live = live && (owner == p.shard);
    }
    return live;
}

// Pixel of a megakernel lane (wavefronts of 8x8 pixels, four per 16x16 tile, RT_TRACE_WAVES per workgroup).
// Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2, so b & 7 names
// the XCD and b >> 3 counts within it.  Macro tiles (RT_MACRO x RT_MACRO tiles) are dealt to the
// XCDs in turn: an XCD works on compact screen regions (its rays share nodes in its L2) while
// every XCD gets an equal share of cheap and expensive parts of the image.  XCD x owns the macro
// tiles x, x + 8, x + 16, ...; the order in which it works through them is their index order, or,
// if p.order is set, the order listed there (lpt_sort_kernel: most expensive first).  macro_out
// receives the macro tile of this wavefront (0xFFFFFFFF: none).
__device__ __forceinline__ bool trace_pixel(const TraceParams &p, bool sharded, int &x, int &y, uint32_t &macro_out) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u;
    uint32_t tx, ty, wave;
    static_assert((RT_MACRO * RT_MACRO * 4) % RT_TRACE_WAVES == 0, "a workgroup must not straddle macro tiles");
    const uint32_t mxn = (p.tiles_x + RT_MACRO - 1) / RT_MACRO, myn = (p.tiles_y + RT_MACRO - 1) / RT_MACRO;
    // j: index of this wavefront among those of its XCD
    const uint32_t j = (b >> 3) * RT_TRACE_WAVES + (threadIdx.x >> 6), per_macro = RT_MACRO * RT_MACRO * 4u;
    const uint32_t k = j / per_macro, r = j % per_macro, xcd = b & 7u;
    macro_out = 0xFFFFFFFFu;
    if (k * 8u + xcd >= mxn * myn) return false;  // k is beyond this XCD's share
    const uint32_t macro = p.order ? p.order[xcd * kLptPerXcd + k] : k * 8u + xcd;
    macro_out = macro;
    const uint32_t t = r >> 2;
    wave = r & 3u;
    tx = (macro % mxn) * RT_MACRO + (t % RT_MACRO);
    ty = (macro / mxn) * RT_MACRO + (t / RT_MACRO);
    if (tx >= p.tiles_x || ty >= p.tiles_y) return false;
    x = (int)(tx * 16u + (wave & 1u) * 8u + (lane & 7u));
    y = (int)(ty * 16u + (wave >> 1) * 8u + (lane >> 3));
    bool live = (x < p.WW) && (y < p.HH);
    if (sharded && p.nshards > 1u) {
        // 64x64 pixel macro tiles dealt round-robin in Morton order
        uint32_t owner = morton2((uint32_t)x >> 6, (uint32_t)y >> 6) % p.nshards;
        live = live && (owner == p.shard);
    }
    return live;
}

// writes a finished pixel-sample: MODE 0 the sRGB-encoded sample image (compute.glsl:673),
// MODE 1 adds the linear radiance into the sum image
template <int MODE>
__device__ __forceinline__ void write_pixel(const TraceParams &p, size_t pix, v3 light) {
    if (MODE == 0) {
        if (p.sample) p.sample[pix] = make_float4(l2s(light.x), l2s(light.y), l2s(light.z), 1.0f);
    } else {
        float4 s = p.sum[pix];
        p.sum[pix] = make_float4(s.x + light.x, s.y + light.y, s.z + light.z, s.w + 1.0f);
    }
}

__device__ __forceinline__ void write_aux_tail(uint32_t *aux, size_t pix, const Path &ps, uint32_t nrays) {
    uint4 *a = reinterpret_cast<uint4 *>(aux + pix * 16);
    a[1] = make_uint4(f2u(ps.D.x), f2u(ps.D.y), f2u(ps.D.z), f2u(ps.O.x));
    a[2] = make_uint4(f2u(ps.O.y), f2u(ps.O.z), f2u(ps.thr.x), f2u(ps.thr.y));
    a[3] = make_uint4(f2u(ps.thr.z), ps.rng, nrays, f2u(ps.ior));
}

// ------------------------------------------------------------------------------------------
// pipeline A — megakernel: one lane = one pixel-sample, the whole of main() (compute.glsl:451-688)
// ------------------------------------------------------------------------------------------
// FAST: the opt-in fast mode -- the same kernel over the library's own tree (rt_fastbvh.cpp): all-axis
// SAH, conservative 8-bit child boxes in 64-byte nodes (4 x 16-byte requests per visit instead of
// 7, and 192 of them fit the workgroup's LDS copy instead of 109).  NOT the reference's tree, so
// results differ where the reference's boxes or visiting order decide which hit is found.
// UNIFIED: nodes and leaves take the same kind of step (traverse_unified); otherwise a node step is
// followed by the intersection of its leaves in steps of their own (node_step).  Which is faster
// depends on the scene: the unified step saves load instructions, the separate steps run less
// code per step (rt_api.cpp times both on the first frames of a scene and keeps the faster one).
template <int MODE, bool AUX, bool STATS, bool FAST, bool UNIFIED>
__device__ __forceinline__ void trace_body(const TraceParams &p) {
    __shared__ uint32_t lds_stack[RT_STACK_LDS * RT_TRACE_BLOCK];
    uint32_t spill[RT_STACK_SPILL];
    const DevScene &sc = p.scene;
    constexpr int TOP = FAST ? RT_FAST_TOP : RT_LDS_TOP;
    // the workgroup's copy of the top of the tree (the first records of the device numbering)
    __shared__ float4 lds_top[RT_LDS_TOP * RT_TOP_STRIDE > RT_FAST_TOP * 4 ? RT_LDS_TOP * RT_TOP_STRIDE : RT_FAST_TOP * 4];
    if (FAST) {
        for (uint32_t k = threadIdx.x; k < RT_FAST_TOP * 4u; k += RT_TRACE_BLOCK)
            lds_top[k] = sc.fnodes[(size_t)min(k >> 2, sc.n_fnodes) * 4u + (k & 3u)];
    } else {
        for (uint32_t k = threadIdx.x; k < RT_LDS_TOP * 7u; k += RT_TRACE_BLOCK) {
            const uint32_t node = k / 7u, part = k % 7u;
            lds_top[node * RT_TOP_STRIDE + part] = sc.nodes[(size_t)min(node, sc.n_nodes) * kNodeVec + part];
        }
    }
    __syncthreads();
    LdsTop top = (LdsTop)lds_top;
    int x, y;
    uint32_t macro;
    // one workgroup in 61 samples the clock it runs at (Counters::clk_cycles / clk_ticks)
    const bool clock_probe = (blockIdx.x % 61u) == 0u;
    const unsigned long long t_begin = (p.cost || clock_probe) ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long r_begin = clock_probe ? __builtin_amdgcn_s_memrealtime() : 0ull;
    bool live = trace_pixel(p, true, x, y, macro);
    uint32_t lane = threadIdx.x & 63u;
    uint32_t *stack = lds_stack + threadIdx.x;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0};
    uint32_t nrays = 0;
    if (live) {
        Path ps;
        primary_ray(sc, p.W, p.H, x, y, p.time, ps);
        const uint32_t nb = (uint32_t)p.bounces;
        uint32_t aux0 = 0xFFFFFFFFu, aux1 = 0u, aux2 = 0u;
        for (uint32_t i = 0; i < nb; i++) {
            // fast mode keeps the reciprocal finite, so that no plane distance becomes NaN
            v3 ID = FAST ? V(fast_rcp(ps.D.x), fast_rcp(ps.D.y), fast_rcp(ps.D.z))
                         : V(1.0f / ps.D.x, 1.0f / ps.D.y, 1.0f / ps.D.z);
            float best = 1e38f;
            uint32_t code = 0xFFFFFFFFu;
            if (UNIFIED && !FAST) {
                traverse_unified<STATS, RT_TRACE_BLOCK, RT_STACK_LDS, TOP>(sc, ps.O, ps.D, ID, best, code, stack, spill, st, top);
            } else {
                int sp = 1;
                stack[0] = 0u;
                if (!FAST) __builtin_amdgcn_s_waitcnt(0x0F70);  // as in traverse_unified: no compiler-known load pending in the loop
                while (sp > 0)
                    node_step<STATS, RT_TRACE_BLOCK, RT_STACK_LDS, TOP, FAST>(sc, ps.O, ps.D, ID, best, code, sp, stack, spill, st, top);
                if (!FAST) code = reference_word(sc, code);
            }
            nrays++;
            bool exists;
            float thit;
            bool alive = bounce<STATS>(sc, ps, best, code, i, exists, thit, st);
            if (AUX && i == 0) {
                aux0 = code;
                aux1 = f2u(thit);
                aux2 = exists ? 1u : 0u;
            }
            if (!alive) break;
        }
        size_t pix = (size_t)y * p.W + x;
        write_pixel<MODE>(p, pix, ps.light);
        if (AUX) {
            reinterpret_cast<uint4 *>(p.aux + pix * 16)[0] = make_uint4(aux0, aux1, aux2, st.nodes);
            write_aux_tail(p.aux, pix, ps, nrays);
        }
    }
    flush_counters<STATS>(p.counters, lane, nrays, st);
    if (p.cost && lane == 0u && macro != 0xFFFFFFFFu)
        atomicAdd(&p.cost[macro], (uint32_t)((__builtin_readcyclecounter() - t_begin) >> 6));
    if (clock_probe && threadIdx.x == 0u) {
        Counters *c = p.counters + (blockIdx.x & (uint32_t)(kCounterStripes - 1));
        atomicAdd(&c->clk_cycles, (unsigned long long)(__builtin_readcyclecounter() - t_begin));
        atomicAdd(&c->clk_ticks, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - r_begin));
    }
}

// One workgroup per XCD: its macro tiles (indices x, x + 8, x + 16, ...) sorted by recorded cost,
// most expensive first, ties by index; the costs are cleared for the next frame.
__global__ __launch_bounds__(1024) void lpt_sort_kernel(uint32_t *cost, uint32_t *order, uint32_t n_macro) {
    __shared__ unsigned long long key[kLptPerXcd];
    const uint32_t xcd = blockIdx.x, t = threadIdx.x;
    const uint32_t n = (n_macro + 7u - xcd) / 8u;  // this XCD's macro tiles
    const uint32_t m = t * 8u + xcd;
    // sorted ascending: a larger cost gives a smaller key, then the smaller index; padding goes last
    key[t] = t < n ? ((unsigned long long)(0xFFFFFFFFu - cost[m]) << 32) | m : ~0ull;
    if (t < n) cost[m] = 0u;
    __syncthreads();
    for (uint32_t size = 2u; size <= kLptPerXcd; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0u; stride >>= 1) {
            const uint32_t partner = t ^ stride;
            if (partner > t) {
                const bool up = (t & size) == 0u;
                const unsigned long long a = key[t], b = key[partner];
                if ((a > b) == up) key[t] = b, key[partner] = a;
            }
            __syncthreads();
        }
    if (t < n) order[xcd * kLptPerXcd + t] = (uint32_t)key[t];
}

void launch_lpt_sort(uint32_t *cost, uint32_t *order, uint32_t n_macro, hipStream_t s) {
    hipLaunchKernelGGL(lpt_sort_kernel, dim3(8), dim3(kLptPerXcd), 0, s, cost, order, n_macro);
}

template <int MODE, bool AUX, bool STATS, bool FAST = false>
__global__ __launch_bounds__(RT_TRACE_BLOCK, RT_MIN_WAVES) void trace_kernel(TraceParams p) {
    trace_body<MODE, AUX, STATS, FAST, true>(p);
}
// the default tree with leaf steps of their own
template <int MODE, bool AUX, bool STATS>
__global__ __launch_bounds__(RT_TRACE_BLOCK, RT_MIN_WAVES) void trace_leafstep_kernel(TraceParams p) {
    trace_body<MODE, AUX, STATS, false, false>(p);
}

// ------------------------------------------------------------------------------------------
// pipeline A' (RT_PIPELINE=regroup) — the megakernel with the live rays of a workgroup regrouped while
// they traverse.
// Rays of one wavefront need very different numbers of node steps, and the wavefront runs until
// its last ray is done.  Here the four wavefronts of a tile stop every RT_REGROUP_STEPS node steps;
// if the rays still traversing would fit into fewer wavefronts, those of the emptiest wavefronts
// ("donors") move into idle lanes of the fullest ones ("keepers") and the donors fall silent for
// the rest of the bounce.  What moves is origin, direction, best hit and where the ray stands in
// its traversal (next item, pending leaves, stack height: 64 bytes per ray through a 4 KB LDS buffer); the LDS stack column stays where it is and is addressed by the ray's home
// thread, the part of a stack that has overflowed into private memory cannot move, so a wavefront
// holding such a ray is always a keeper.  A ray's own sequence of node visits and intersection
// tests is untouched: results are those of trace_kernel bit for bit.  Hit distance and primitive
// return to the home lane through the (then empty) stack column; everything outside traversal
// runs in the home lane.
// ------------------------------------------------------------------------------------------
#ifndef RT_REGROUP_STEPS
#define RT_REGROUP_STEPS 8
#endif
#define RT_RG_TOP 69  // leading nodes in LDS: what is left of 20 KB next to the stack and the transfer buffer

template <int MODE, bool AUX, bool STATS>
__global__ __launch_bounds__(RT_BLOCK, RT_MIN_WAVES) void trace_regroup_kernel(TraceParams p) {
    __shared__ uint32_t lds_stack[RT_STACK_LDS * RT_BLOCK];
    __shared__ float4 lds_top[RT_RG_TOP * RT_TOP_STRIDE];
    __shared__ float4 xfer[4 * 64];              // a moving ray: O.xyz best | D.xyz primitive | its next item, three pending
                                                 // leaves | the fourth, the pending mask, stack height, home thread
    __shared__ uint32_t xvisits[STATS ? 64 : 1];  //   node visits of this ray so far (instrumented builds)
    __shared__ unsigned long long rg_mask[2][4];  // per round parity and wavefront: lanes with a ray still traversing
    __shared__ uint32_t rg_pin[2][4];             //   the wavefront holds a ray whose stack reaches into private memory
    uint32_t spill[RT_STACK_SPILL];
    const DevScene &sc = p.scene;
    for (uint32_t k = threadIdx.x; k < RT_RG_TOP * 7u; k += RT_BLOCK) {
        const uint32_t node = k / 7u, part = k % 7u;
        lds_top[node * RT_TOP_STRIDE + part] = sc.nodes[(size_t)min(node, sc.n_nodes) * kNodeVec + part];
    }
    __syncthreads();
    LdsTop top = (LdsTop)lds_top;
    int x, y;
    uint32_t macro_unused;
    const bool live = trace_pixel(p, true, x, y, macro_unused);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0};
    uint32_t nrays = 0, path_visits = 0;
    Path ps;
    if (live) primary_ray(sc, p.W, p.H, x, y, p.time, ps);
    const uint32_t nb = (uint32_t)p.bounces;
    bool path_alive = live;
    uint32_t aux0 = 0xFFFFFFFFu, aux1 = 0u, aux2 = 0u;
    uint32_t round = 0;
    for (uint32_t i = 0; i < nb; i++) {
        // the ray this lane traces: first its own
        v3 O = ps.O, D = ps.D;
        v3 ID = V(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
        float best = 1e38f;
        uint32_t code = 0xFFFFFFFFu, slot = threadIdx.x, visits = 0;
        Walk w;
        w.cur = 0u, w.ll = {0u, 0u, 0u, 0u, 0}, w.sp = 0, w.have = false;
        bool has = path_alive;
        if (has) walk_begin<STATS>(w, st);
        // bit 1 of what a wavefront publishes next to its mask: some path of it was alive at this bounce
        const uint32_t wave_has_paths = __ballot(path_alive) != 0ull ? 2u : 0u;
        bool any_path = true;
        for (;; round++) {
            for (int k = 0; k < RT_REGROUP_STEPS; k++) {
                if (__ballot(w.have) == 0ull) break;
                if (w.have) {
                    const uint32_t before = st.nodes;
                    walk_step<STATS, RT_BLOCK, RT_STACK_LDS, RT_RG_TOP>(sc, O, D, ID, best, code, w, lds_stack + slot, spill, st, top);
                    if (STATS) visits += st.nodes - before;
                }
            }
            // a finished ray leaves its result in its home column
            if (has && !w.have) {
                lds_stack[slot] = f2u(best);
                lds_stack[RT_BLOCK + slot] = reference_word(sc, code);
                if (STATS) lds_stack[2 * RT_BLOCK + slot] = visits;
                has = false;
            }
            const unsigned long long mine = __ballot(has);
            const bool pinned = __ballot(has && w.sp > RT_STACK_LDS) != 0ull;
            const uint32_t par = round & 1u;
            if (lane == 0) {
                rg_mask[par][wave] = mine;
                rg_pin[par][wave] = (pinned ? 1u : 0u) | wave_has_paths;
            }
            __syncthreads();
            uint32_t cnt[4], key[4], total = 0, npin = 0, paths = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const unsigned long long m = rg_mask[par][w];
                const uint32_t pn = rg_pin[par][w];
                cnt[w] = (uint32_t)__builtin_amdgcn_readfirstlane((int)__popcll(m));
                const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)pn);
                const uint32_t pin_w = flags & 1u;
                paths |= flags & 2u;
                key[w] = cnt[w] + (pin_w << 8);
                total += cnt[w];
                npin += pin_w;
            }
            if (total == 0u) {
                any_path = paths != 0u;
                break;
            }
            // keepers: the wavefronts with pinned rays, then the fullest ones, as many as the rays need
            const uint32_t nkeep = max((total + 63u) / 64u, npin);
            bool keep[4];
            uint32_t movers = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t rank = 0;
#pragma unroll
                for (int v = 0; v < 4; v++) rank += (key[v] > key[w] || (key[v] == key[w] && v < w)) ? 1u : 0u;
                keep[w] = rank < nkeep;
                movers += keep[w] ? 0u : cnt[w];
            }
            // nothing to move, or (only with pinned wavefronts among the keepers) more than the buffer holds
            if (movers == 0u || movers > 64u) continue;
            uint32_t mover_base = 0, hole_base = 0;
            bool keeper = false;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if ((uint32_t)w < wave) {
                    mover_base += keep[w] ? 0u : cnt[w];
                    hole_base += keep[w] ? 64u - cnt[w] : 0u;
                }
                if ((uint32_t)w == wave) keeper = keep[w];
            }
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
            if (!keeper && has) {
                const uint32_t m = mover_base + below;
                xfer[4 * m] = make_float4(O.x, O.y, O.z, best);
                xfer[4 * m + 1] = make_float4(D.x, D.y, D.z, u2f(code));
                xfer[4 * m + 2] = make_float4(u2f(w.cur), u2f(w.ll.v0), u2f(w.ll.v1), u2f(w.ll.v2));
                xfer[4 * m + 3] = make_float4(u2f(w.ll.v3), u2f(w.ll.pend), u2f((uint32_t)w.sp), u2f(slot));
                if (STATS) xvisits[m] = visits;
                has = false;
                w.have = false;
            }
            __syncthreads();
            if (keeper && !has) {
                const uint32_t h = hole_base + (lane - below);
                if (h < movers) {
                    const float4 a = xfer[4 * h], b = xfer[4 * h + 1], c2 = xfer[4 * h + 2], d2 = xfer[4 * h + 3];
                    O = V(a.x, a.y, a.z), D = V(b.x, b.y, b.z);
                    ID = V(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
                    best = a.w, code = f2u(b.w);
                    w.cur = f2u(c2.x), w.ll.v0 = f2u(c2.y), w.ll.v1 = f2u(c2.z), w.ll.v2 = f2u(c2.w);
                    w.ll.v3 = f2u(d2.x), w.ll.pend = f2u(d2.y), w.sp = (int)f2u(d2.z), slot = f2u(d2.w);
                    w.have = true;
                    if (STATS) visits = xvisits[h];
                    has = true;
                }
            }
        }
        round++;
        if (!any_path) break;  // no path of this tile is alive any more
        if (path_alive) {
            best = u2f(lds_stack[threadIdx.x]);
            code = lds_stack[RT_BLOCK + threadIdx.x];
            if (STATS) path_visits += lds_stack[2 * RT_BLOCK + threadIdx.x];
            nrays++;
            bool exists;
            float thit;
            path_alive = bounce<STATS>(sc, ps, best, code, i, exists, thit, st);
            if (AUX && i == 0) {
                aux0 = code;
                aux1 = f2u(thit);
                aux2 = exists ? 1u : 0u;
            }
        }
    }
    if (live) {
        size_t pix = (size_t)y * p.W + x;
        write_pixel<MODE>(p, pix, ps.light);
        if (AUX) {
            reinterpret_cast<uint4 *>(p.aux + pix * 16)[0] = make_uint4(aux0, aux1, aux2, path_visits);
            write_aux_tail(p.aux, pix, ps, nrays);
        }
    }
    flush_counters<STATS>(p.counters, lane, nrays, st);
}

void launch_trace_regroup(const TraceParams &p, int mode, bool aux, bool stats, hipStream_t s) {
    if (p.ntiles == 0) return;
    static_assert(RT_TRACE_WAVES == 4, "trace_pixel must map a workgroup to the four wavefronts of one tile");
    const uint32_t mx = (p.tiles_x + RT_MACRO - 1) / RT_MACRO, my = (p.tiles_y + RT_MACRO - 1) / RT_MACRO;
    const uint32_t groups = ((mx * my + 7u) / 8u) * 8u * (RT_MACRO * RT_MACRO);
    dim3 grid(groups), block(RT_BLOCK);
#define L(M, A, S) hipLaunchKernelGGL((trace_regroup_kernel<M, A, S>), grid, block, 0, s, p)
    if (mode == 0) {
        if (aux && stats) L(0, true, true);
        else if (aux) L(0, true, false);
        else if (stats) L(0, false, true);
        else L(0, false, false);
    } else {
        if (stats) L(1, false, true);
        else L(1, false, false);
    }
#undef L
}

void launch_trace_fast(const TraceParams &p, int mode, bool aux, bool stats, hipStream_t s) {
    if (p.ntiles == 0) return;
    const uint32_t mx = (p.tiles_x + RT_MACRO - 1) / RT_MACRO, my = (p.tiles_y + RT_MACRO - 1) / RT_MACRO;
    const uint32_t groups = ((mx * my + 7u) / 8u) * 8u * (RT_MACRO * RT_MACRO * 4u) / RT_TRACE_WAVES;
    dim3 grid(groups), block(RT_TRACE_BLOCK);
#define L(M, A, S) hipLaunchKernelGGL((trace_kernel<M, A, S, true>), grid, block, 0, s, p)
    if (mode == 0) {
        if (aux && stats) L(0, true, true);
        else if (aux) L(0, true, false);
        else if (stats) L(0, false, true);
        else L(0, false, false);
    } else {
        if (stats) L(1, false, true);
        else L(1, false, false);
    }
#undef L
}

void launch_trace(const TraceParams &p, int mode, bool aux, bool stats, bool leaf_steps, hipStream_t s) {
    if (p.ntiles == 0) return;  // image smaller than one dispatch group: the reference dispatches nothing
    // 8 equal runs of workgroups, one per XCD; surplus workgroups exit at once
    const uint32_t mx = (p.tiles_x + RT_MACRO - 1) / RT_MACRO, my = (p.tiles_y + RT_MACRO - 1) / RT_MACRO;
    const uint32_t groups = ((mx * my + 7u) / 8u) * 8u * (RT_MACRO * RT_MACRO * 4u) / RT_TRACE_WAVES;
    dim3 grid(groups), block(RT_TRACE_BLOCK);
#define L(M, A, S)                                                                                  \
    do {                                                                                            \
        if (leaf_steps) hipLaunchKernelGGL((trace_leafstep_kernel<M, A, S>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((trace_kernel<M, A, S>), grid, block, 0, s, p);                     \
    } while (0)
    if (mode == 0) {
        if (aux && stats) L(0, true, true);
        else if (aux) L(0, true, false);
        else if (stats) L(0, false, true);
        else L(0, false, false);
    } else {
        if (stats) L(1, false, true);
        else L(1, false, false);
    }
#undef L
}

// ------------------------------------------------------------------------------------------
// pipeline B — wavefront: rays live in a queue in HBM; per bounce one persistent EXTEND kernel
// (traversal only, lanes refilled from the queue as they finish) and one SHADE kernel (the rest
// of the loop body, rewrites the ray in place for the next bounce).  Same arithmetic per path as
// the megakernel, only the scheduling differs.  Every pixel-lane of the tile grid owns one queue
// slot for the whole frame (slot = tile * 256 + lane-in-tile), so no stage needs an atomic; a
// slot whose path has ended carries the pixel index 0xFFFFFFFF.
//   ray record (32 B):   [0] O.xyz, pixel index   [1] D.xyz, unused
//   hit record ( 8 B):   distance, metadata word of the primitive
//   path record (32 B):  [0] throughput.rgb, rng   [1] light.rgb, ior
// ------------------------------------------------------------------------------------------
#define RT_DEAD 0xFFFFFFFFu

template <int MODE, bool AUX>
__global__ __launch_bounds__(RT_BLOCK) void generate_kernel(TraceParams p, WaveQueues q) {
    const DevScene &sc = p.scene;
    int x, y;
    bool live = tile_pixel(p, true, x, y);
    const uint32_t nb = (uint32_t)p.bounces;
    // slot of this lane: its tile (after the XCD remap) times 256 plus its position in the tile
    const uint32_t per = (p.ntiles + 7u) / 8u;
    const uint32_t tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    const size_t slot = (size_t)tile * RT_BLOCK + threadIdx.x;
    Path ps;
    uint32_t pix = RT_DEAD;
    if (live) {
        primary_ray(sc, p.W, p.H, x, y, p.time, ps);
        pix = (uint32_t)y * (uint32_t)p.W + (uint32_t)x;
        if (AUX) {
            reinterpret_cast<uint4 *>(p.aux + (size_t)pix * 16)[0] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
            write_aux_tail(p.aux, pix, ps, 0u);
        }
        if (nb == 0u) {
            write_pixel<MODE>(p, pix, ps.light);
            pix = RT_DEAD;
        }
    }
    if (pix != RT_DEAD) {
        q.rays[0][slot * 2] = make_float4(ps.O.x, ps.O.y, ps.O.z, u2f(pix));
        q.rays[0][slot * 2 + 1] = make_float4(ps.D.x, ps.D.y, ps.D.z, 0.f);
        q.paths[slot * 2] = make_float4(ps.thr.x, ps.thr.y, ps.thr.z, u2f(ps.rng));
        q.paths[slot * 2 + 1] = make_float4(ps.light.x, ps.light.y, ps.light.z, ps.ior);
    } else {
        q.rays[0][slot * 2] = make_float4(0.f, 0.f, 0.f, u2f(RT_DEAD));
    }
}

// Persistent traversal kernel.  Workgroups of four independent wavefronts (one-wavefront
// workgroups cannot fill a CU: at most 8 workgroups are resident per CU, i.e. 2 waves per SIMD).
// The queue is cut in 8 equal regions, one per XCD (the tiles whose rays a region holds were
// generated by workgroups of the same XCD under round-robin placement, so their nodes are in that
// XCD's L2); within its XCD's region wavefront w of the Wx resident there owns the 64-slot chunks
// w, w + Wx, w + 2 Wx, ...: a static interleaved partition, so nothing is claimed through memory.
// Lanes whose ray is done retire their result and are refilled in batches; every lane with work
// performs one step (a node visit or a leaf intersection: walk_step) per iteration.  A wavefront
// ends when its chunks are used up and every lane is idle.  The workgroup shares one LDS copy of
// the top of the tree.
template <bool STATS>
__global__ __launch_bounds__(RT_BLOCK, RT_EXTEND_MIN_WAVES) void extend_kernel(DevScene sc, WaveQueues q, uint32_t n_slots,
                                                                              uint32_t refill_threshold, Counters *counters,
                                                                              uint32_t *aux) {
    __shared__ uint32_t lds_stack[RT_EXT_STACK_LDS * RT_BLOCK];
    uint32_t spill[RT_STACK_SPILL];
#if RT_LDS_TOP > 0
    __shared__ float4 lds_top[RT_LDS_TOP * RT_TOP_STRIDE];
    for (uint32_t k = threadIdx.x; k < RT_LDS_TOP * 7u; k += RT_BLOCK) {
        const uint32_t node = k / 7u, part = k % 7u;
        lds_top[node * RT_TOP_STRIDE + part] = sc.nodes[(size_t)min(node, sc.n_nodes) * kNodeVec + part];
    }
    __syncthreads();  // the only barrier: the wavefronts are independent from here on
    LdsTop top = (LdsTop)lds_top;
#else
    LdsTop top = nullptr;
#endif
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *stack = lds_stack + wave * (RT_EXT_STACK_LDS * 64u) + lane;
    const float4 *__restrict__ rays = q.rays[0];
    float2 *__restrict__ hits = q.hits;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0};
    uint32_t nrays_done = 0;
    // wave-uniform cursor over this wavefront's chunks
    const uint32_t n_chunks = (n_slots + 63u) / 64u, per_xcd = (n_chunks + 7u) / 8u;
    const uint32_t xcd = blockIdx.x & 7u, waves_x = (gridDim.x >> 3) * (RT_BLOCK / 64u);
    const uint32_t region_end = min((xcd + 1u) * per_xcd, n_chunks);
    uint32_t chunk = xcd * per_xcd + (blockIdx.x >> 3) * (RT_BLOCK / 64u) + wave;
    uint32_t cur = chunk * 64u, end = min(cur + 64u, n_slots);
    bool exhausted = chunk >= region_end;
    int ray = -1;
    v3 O = V(0, 0, 0), D = V(0, 0, 0), ID = V(0, 0, 0);
    float best = 1e38f;
    uint32_t code = 0xFFFFFFFFu;
    uint32_t nodes_this_ray = 0, pix_this_ray = 0;
    Walk w;
    w.cur = 0u, w.ll = {0u, 0u, 0u, 0u, 0}, w.sp = 0, w.have = false;
    for (;;) {
        bool idle = !w.have;
        unsigned long long idle_mask = __ballot(idle);
        uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (n_idle >= refill_threshold || n_idle == 64u) {
            if (idle && ray >= 0) {
                hits[ray] = make_float2(best, u2f(reference_word(sc, code)));
                if (STATS && aux) atomicAdd(aux + (size_t)pix_this_ray * 16 + 3, nodes_this_ray);
                nrays_done++;
                ray = -1;
            }
            if (!exhausted) {
                // hand the idle lanes consecutive slots from the current chunk(s)
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                uint32_t given = 0;  // idle lanes served so far (wave-uniform)
                while (given < n_idle && !exhausted) {
                    uint32_t take = min(n_idle - given, end - cur);
                    if (idle && rank >= given && rank < given + take) {
                        uint32_t slot = cur + (rank - given);
                        float4 r0 = rays[(size_t)slot * 2];
                        if (f2u(r0.w) != RT_DEAD) {
                            float4 r1 = rays[(size_t)slot * 2 + 1];
                            ray = (int)slot;
                            pix_this_ray = f2u(r0.w);
                            O = V(r0.x, r0.y, r0.z);
                            D = V(r1.x, r1.y, r1.z);
                            ID = V(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
                            best = 1e38f;
                            code = 0xFFFFFFFFu;
                            walk_begin<STATS>(w, st);
                            nodes_this_ray = 0;
                        }
                    }
                    cur += take;
                    given += take;
                    if (cur >= end) {
                        chunk += waves_x;
                        cur = chunk * 64u;
                        end = min(cur + 64u, n_slots);
                        if (chunk >= region_end) exhausted = true;
                    }
                }
            }
            if (exhausted && __ballot(w.have) == 0ull) break;
        }
        if (w.have) {
            const uint32_t visits_before = st.nodes;
            walk_step<STATS, 64, RT_EXT_STACK_LDS, RT_LDS_TOP>(sc, O, D, ID, best, code, w, stack, spill, st, top);
            if (STATS) nodes_this_ray += st.nodes - visits_before;
        }
    }
    flush_counters<STATS>(counters, lane, nrays_done, st);
}

// The rest of the loop body for every live slot, one lane per slot; the surviving ray replaces
// the old one.  BIN = false: in place.  BIN = true (RT_WAVE_BIN=1): the 256 slots of the block are
// rewritten grouped by the octant of the new ray's direction (a counting sort over the block: wave
// ballots, one barrier), ended paths last, so that the traversal kernel's 64-slot chunks hold rays
// that point the same way and no gaps; records move together, and the pixel index travels in the
// ray record, so nothing else needs to know where a path sits.
template <int MODE, bool AUX, bool STATS, bool BIN>
__global__ __launch_bounds__(RT_BLOCK) void shade_kernel(TraceParams p, WaveQueues q, uint32_t n_slots,
                                                        uint32_t bounce_index) {
    const DevScene &sc = p.scene;
    const size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nb = (uint32_t)p.bounces;
    LaneStats st = {0, 0, 0, 0, 0, 0, 0};
    Path ps;
    uint32_t pix = RT_DEAD;
    bool alive = false, was_live = false;
    if (i < n_slots) {
        float4 r0 = q.rays[0][i * 2];
        pix = f2u(r0.w);
        if (pix != RT_DEAD) {
            was_live = true;
            float4 r1 = q.rays[0][i * 2 + 1];
            float2 h = q.hits[i];
            float4 s0 = q.paths[i * 2], s1 = q.paths[i * 2 + 1];
            ps.O = V(r0.x, r0.y, r0.z);
            ps.D = V(r1.x, r1.y, r1.z);
            ps.thr = V(s0.x, s0.y, s0.z);
            ps.rng = f2u(s0.w);
            ps.light = V(s1.x, s1.y, s1.z);
            ps.ior = s1.w;
            bool exists;
            float thit;
            alive = bounce<STATS>(sc, ps, h.x, f2u(h.y), bounce_index, exists, thit, st);
            if (AUX && bounce_index == 0u) {
                uint32_t *a = p.aux + (size_t)pix * 16;
                a[0] = f2u(h.y);
                a[1] = f2u(thit);
                a[2] = exists ? 1u : 0u;
            }
            alive = alive && (bounce_index + 1u < nb);
            if (!alive) {
                write_pixel<MODE>(p, pix, ps.light);
                if (AUX) write_aux_tail(p.aux, pix, ps, bounce_index + 1u);
            }
        }
    }
    size_t dst = i;
    if (BIN) {
        __shared__ uint32_t wave_count[RT_BLOCK / 64][9];
        const uint32_t wave = threadIdx.x >> 6;
        const uint32_t key = alive ? ((ps.D.x < 0.0f ? 1u : 0u) | (ps.D.y < 0.0f ? 2u : 0u) | (ps.D.z < 0.0f ? 4u : 0u)) : 8u;
        uint32_t rank_in_wave = 0;
#pragma unroll
        for (uint32_t k = 0; k < 9u; k++) {
            const unsigned long long m = __ballot(key == k);
            if (key == k)
                rank_in_wave = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (lane == 0u) wave_count[wave][k] = (uint32_t)__popcll(m);
        }
        __syncthreads();  // every record of the block has been read by now: the block's slots may be rewritten
        uint32_t base = 0;
        for (uint32_t k = 0; k < 9u; k++)
            for (uint32_t w = 0; w < RT_BLOCK / 64u; w++)
                if (k < key || (k == key && w < wave)) base += wave_count[w][k];
        dst = (size_t)blockIdx.x * RT_BLOCK + base + rank_in_wave;
    }
    if (i < n_slots) {
        if (alive) {
            q.rays[0][dst * 2] = make_float4(ps.O.x, ps.O.y, ps.O.z, u2f(pix));
            q.rays[0][dst * 2 + 1] = make_float4(ps.D.x, ps.D.y, ps.D.z, 0.f);
            q.paths[dst * 2] = make_float4(ps.thr.x, ps.thr.y, ps.thr.z, u2f(ps.rng));
            q.paths[dst * 2 + 1] = make_float4(ps.light.x, ps.light.y, ps.light.z, ps.ior);
        } else if (was_live || BIN) {
            q.rays[0][dst * 2] = make_float4(0.f, 0.f, 0.f, u2f(RT_DEAD));
        }
    }
    if (STATS) flush_counters<true>(p.counters, lane, 0u, st);
}

void launch_wavefront_frame(const TraceParams &p, const WaveQueues &q, int mode, bool aux, bool stats,
                            uint32_t extend_waves, uint32_t refill_threshold, bool bin_by_octant, hipStream_t s) {
    uint32_t per = (p.ntiles + 7u) / 8u;
    if (per == 0) return;
    const uint32_t nb_raw = (uint32_t)p.bounces;
    const uint32_t nb = nb_raw > 65u ? 65u : nb_raw;  // a path never survives bounce index 64
    dim3 tgrid(per * 8u), block(RT_BLOCK);
    const uint32_t n_slots = per * 8u * RT_BLOCK;
#define G(M, A) hipLaunchKernelGGL((generate_kernel<M, A>), tgrid, block, 0, s, p, q)
    if (mode == 0) {
        if (aux) G(0, true);
        else G(0, false);
    } else {
        G(1, false);
    }
#undef G
    // workgroups of four wavefronts in a multiple of 8 (one share per XCD), no more wavefronts than chunks
    const uint32_t n_chunks = n_slots / 64u;
    uint32_t egroups = ((extend_waves < n_chunks ? extend_waves : n_chunks) / (RT_BLOCK / 64u) + 7u) / 8u * 8u;
    if (egroups < 8u) egroups = 8u;
    for (uint32_t b = 0; b < nb; b++) {
        if (stats)
            hipLaunchKernelGGL((extend_kernel<true>), dim3(egroups), dim3(RT_BLOCK), 0, s, p.scene, q, n_slots,
                               refill_threshold, p.counters, aux ? p.aux : nullptr);
        else
            hipLaunchKernelGGL((extend_kernel<false>), dim3(egroups), dim3(RT_BLOCK), 0, s, p.scene, q, n_slots,
                               refill_threshold, p.counters, (uint32_t *)nullptr);
#define S(M, A, T)                                                                                       \
    do {                                                                                                 \
        if (bin_by_octant) hipLaunchKernelGGL((shade_kernel<M, A, T, true>), tgrid, block, 0, s, p, q, n_slots, b);  \
        else hipLaunchKernelGGL((shade_kernel<M, A, T, false>), tgrid, block, 0, s, p, q, n_slots, b);   \
    } while (0)
        if (mode == 0) {
            if (aux && stats) S(0, true, true);
            else if (aux) S(0, true, false);
            else if (stats) S(0, false, true);
            else S(0, false, false);
        } else {
            if (stats) S(1, false, true);
            else S(1, false, false);
        }
#undef S
    }
}

// ------------------------------------------------------------------------------------------
// accumulate pass (shaders/screen_fragment.glsl:94-108) and sum finalisation
// ------------------------------------------------------------------------------------------
__global__ void accumulate_kernel(const float4 *__restrict__ sample, float4 *__restrict__ acc, size_t n,
                                  float frames) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w = 1.0f / (frames + 1.0f);
    float4 s = sample[i], a = acc[i];
    float ar = s2l(a.x), ag = s2l(a.y), ab = s2l(a.z);
    float sr = s2l(s.x), sg = s2l(s.y), sb = s2l(s.z);
    acc[i] = make_float4(l2s(ar + w * (sr - ar)), l2s(ag + w * (sg - ag)), l2s(ab + w * (sb - ab)), 1.0f);
}

__global__ void finalize_sum_kernel(const float4 *__restrict__ sum, float4 *__restrict__ acc, size_t n, float spp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = sum[i];
    acc[i] = make_float4(l2s(s.x / spp), l2s(s.y / spp), l2s(s.z / spp), 1.0f);
}

// target += part; part = 0
__global__ void fold_sum_kernel(float4 *__restrict__ target, float4 *__restrict__ part, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 t = target[i], s = part[i];
    target[i] = make_float4(t.x + s.x, t.y + s.y, t.z + s.z, t.w + s.w);
    part[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_fold_sum(float4 *target, float4 *part, size_t npixels, hipStream_t s) {
    hipLaunchKernelGGL(fold_sum_kernel, dim3((unsigned)((npixels + 255) / 256)), dim3(256), 0, s, target, part, npixels);
}

void launch_accumulate(const float4 *sample, float4 *acc, size_t npixels, float frames, hipStream_t s) {
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((npixels + 255) / 256)), dim3(256), 0, s, sample, acc,
                       npixels, frames);
}
void launch_finalize_sum(const float4 *sum, float4 *acc, size_t npixels, uint32_t spp, hipStream_t s) {
    hipLaunchKernelGGL(finalize_sum_kernel, dim3((unsigned)((npixels + 255) / 256)), dim3(256), 0, s, sum, acc,
                       npixels, (float)spp);
}

}  // namespace rt
