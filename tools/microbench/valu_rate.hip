// Issue rate of plain and packed fp32 VALU instructions on one SIMD as a function of the number of
// resident wavefronts.  Each wavefront runs a long stream of independent multiplies.
// Cycles are shader-clock cycles as the kernel itself counts them (s_memtime), and the clock the
// chip actually holds is reported next to them (s_memtime / s_memrealtime, the latter at 100 MHz):
// dividing wall time by the nominal clock instead overstates the cycles per instruction whenever
// the chip runs below its nominal clock under load.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int PACKED>
__global__ __launch_bounds__(256) void stream_kernel(float *out, int iters, long long *cycles) {
    float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
          a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m};
    long long r0 = __builtin_amdgcn_s_memrealtime();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (PACKED) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    float r = PACKED ? (p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y) : (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        cycles[2 * w] = t1 - t0;
        cycles[2 * w + 1] = r1 - r0;
    }
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float *out;
    long long *cyc;
    const int max_blocks = cus * 4 * 8;  // in wavefronts
    CHECK(hipMalloc(&out, sizeof(float) * 64 * max_blocks));
    CHECK(hipMalloc(&cyc, sizeof(long long) * 2 * max_blocks));
    std::vector<long long> hc(2 * (size_t)max_blocks);
    const int iters = 20000;  // 32 instructions per iteration in both variants
    printf("CUs %d, clock %d kHz\n", cus, prop.clockRate);
    for (int packed = 0; packed < 2; packed++) {
        for (int wps = 1; wps <= 8; wps *= 2) {
            // workgroups of four wavefronts (one per SIMD), wps of them per CU.  One-wavefront workgroups
            // would not do: at most 8 workgroups are resident per CU, so they never exceed 2 waves per SIMD.
            int blocks = cus * 4 * wps;  // wavefronts launched
            const int groups = cus * wps;
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                if (packed) hipLaunchKernelGGL(stream_kernel<1>, dim3(groups), dim3(256), 0, 0, out, iters, cyc);
                else hipLaunchKernelGGL(stream_kernel<0>, dim3(groups), dim3(256), 0, 0, out, iters, cyc);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
            }
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            double insts_per_simd = (double)iters * 32 * wps;
            CHECK(hipMemcpy(hc.data(), cyc, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost));
            // median wave: its own cycle count and the clock it ran at
            std::vector<long long> cs, rs;
            for (int b = 0; b < blocks; b++) cs.push_back(hc[2 * b]), rs.push_back(hc[2 * b + 1]);
            std::nth_element(cs.begin(), cs.begin() + blocks / 2, cs.end());
            std::nth_element(rs.begin(), rs.begin() + blocks / 2, rs.end());
            const double wave_cycles = (double)cs[blocks / 2], ghz = wave_cycles / ((double)rs[blocks / 2] * 10.0);
            // wps waves share a SIMD for the whole run when the chip is full (wps = 8) or when the
            // dispatcher spreads them evenly; the in-kernel figure needs no such assumption at wps = 8
            printf("%s waves/SIMD %d: %.3f ms wall; median wave %.0f cycles at %.3f GHz -> %.2f cycles per wave-instruction "
                   "per SIMD in-kernel, %.2f from wall time at that clock (%.2f at the nominal %.1f GHz)\n",
                   packed ? "v_pk_mul_f32" : "v_mul_f32   ", wps, ms, wave_cycles, ghz, wave_cycles / ((double)iters * 32 * wps),
                   ms * 1e-3 * ghz * 1e9 / insts_per_simd, ms * 1e-3 * prop.clockRate * 1e3 / insts_per_simd,
                   prop.clockRate * 1e-6);
        }
    }
    return 0;
}
