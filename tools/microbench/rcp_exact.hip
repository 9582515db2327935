// Is r = fma(fma(-x, r0, 1), r0, r0) with r0 = v_rcp_f32(x) equal to the correctly rounded 1.0f / x?
// Exhaustive over all 2^32 bit patterns; reports mismatches by magnitude class.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/microbench/rcp_exact.hip -o build/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void check(unsigned long long *mism_in_range, unsigned long long *mism_all, uint32_t *examples) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long a = 0, b = 0;
    for (uint64_t u = tid; u < (1ull << 32); u += stride) {
        float x = __uint_as_float((uint32_t)u);
        float exact = 1.0f / x;
        float r0 = __builtin_amdgcn_rcpf(x);
        float e = __builtin_fmaf(-x, r0, 1.0f);
        float r1 = __builtin_fmaf(e, r0, r0);
        bool same = (__float_as_uint(exact) == __float_as_uint(r1)) || (exact != exact && r1 != r1);
        if (!same) {
            b++;
            float ax = fabsf(x);
            if (ax >= 1e-6f && ax <= 1e30f) {
                unsigned long long k = atomicAdd(mism_in_range, 1ull);
                if (k < 16) examples[k] = (uint32_t)u;
                a++;
            }
        }
    }
    atomicAdd(mism_all, b);
}

int main() {
    unsigned long long *d;
    uint32_t *ex;
    CHECK(hipMalloc(&d, 16));
    CHECK(hipMalloc(&ex, 64));
    CHECK(hipMemset(d, 0, 16));
    CHECK(hipMemset(ex, 0, 64));
    hipLaunchKernelGGL(check, dim3(256 * 16), dim3(256), 0, 0, d, d + 1, ex);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[2];
    uint32_t hex[16];
    CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hex, ex, 64, hipMemcpyDeviceToHost));
    printf("mismatches with 1e-6 <= |x| <= 1e30: %llu; over all bit patterns: %llu\n", h[0], h[1]);
    for (int i = 0; i < 16 && i < (int)h[0]; i++) {
        float x;
        memcpy(&x, &hex[i], 4);
        printf("  x = %a (%08x)\n", x, hex[i]);
    }
    return 0;
}
