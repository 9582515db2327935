// Calibrates rocprofv3's FETCH_SIZE for this library's access pattern: every lane reads 7 x 16 B
// of one random 128-byte record (what a node visit does) from a table far larger than L2 and the
// Infinity Cache, so nearly every record is a miss that must come from HBM.  Known bytes:
// lanes x 128 B if a whole line is fetched per miss (lanes x 112 B touched).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o build/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE TCC_MISS_sum -- build/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void gather_kernel(const float4 *__restrict__ table, uint32_t records, float *out, uint32_t seed) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = i * 747796405u + seed;
    s = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    s = (s >> 22u) ^ s;
    const float4 *p = table + (size_t)(s % records) * 8;
    float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6];
    out[i] = a.x + b.x + c.x + d.x + e.x + f.x + g.x;
}

int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB table
    const uint32_t records = (uint32_t)(bytes / 128);
    const uint32_t lanes = 1u << 22;       // 4 Mi lanes -> 4 Mi record reads per launch
    float4 *table;
    float *out;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipMalloc(&out, lanes * sizeof(float)));
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL(gather_kernel, dim3(lanes / 256), dim3(256), 0, 0, table, records, out, 12345u + rep * 977u);
        CHECK(hipDeviceSynchronize());
    }
    printf("per launch: %u record reads, %.1f MB at 128 B per record, %.1f MB at 64 B\n", lanes, lanes * 128.0 / 1e6,
           lanes * 64.0 / 1e6);
    return 0;
}
