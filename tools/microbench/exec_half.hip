// Does a VALU instruction whose EXEC mask covers only one 32-lane half (or fewer lanes) issue faster
// than one with all 64 lanes active?  Same instruction stream, different EXEC.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/exec_half.hip -o build/exec_half && build/exec_half
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k(float *out, int iters, unsigned long long mask) {
    float a0 = threadIdx.x * 0.001f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
          a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0000001f;
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        }
    }
    asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, wps = 8, blocks = cus * 4 * wps, iters = 20000;
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 64 * blocks));
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 0xFull, 0x0000000100000001ull, 0xFFFF0000FFFF0000ull};
    const char *names[] = {"all 64 lanes", "lower 32 lanes", "lower 16 lanes", "lower 4 lanes", "lanes 0 and 32", "lanes 16-31,48-63"};
    for (int v = 0; v < 6; v++) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, iters, masks[v]);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        double cyc = ms * 1e-3 * prop.clockRate * 1e3 / ((double)iters * 32 * wps);
        printf("%-20s %.3f ms  %.2f nominal cycles per wave-instruction per SIMD (8 waves/SIMD)\n", names[v], ms, cyc);
    }
    return 0;
}
