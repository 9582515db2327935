// Microbenchmark: throughput of per-lane 16-byte gathers from an L1/L2-resident table on gfx950,
// as a function of how the 64 lanes of a wave spread over cache lines.  Guides the node layout.
//   pattern 0: every lane a different random 128-B line (16 B each)
//   pattern 1: groups of 4 lanes read the 4 x 16 B quarters of one random 64-B chunk
//   pattern 2: groups of 8 lanes read one random 128-B line (16 B each)
//   pattern 3: all lanes the same address
//   pattern 4: every lane a different random line, 4 dependent-free loads covering 64 contiguous B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PATTERN>
__global__ void gather(const float4* __restrict__ table, uint32_t nlines, int iters, float4* out, uint32_t seed) {
    uint32_t lane = threadIdx.x & 63u;
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + seed;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u;
        uint32_t r = s >> 8;
        size_t idx;  // in float4 units; a 128-B line = 8 float4
        if (PATTERN == 0) idx = (size_t)(r % nlines) * 8 + (r >> 20 & 7);
        else if (PATTERN == 1) { uint32_t g = __shfl(r, lane & ~3u); idx = (size_t)(g % nlines) * 8 + ((g >> 20 & 1) * 4) + (lane & 3u); }
        else if (PATTERN == 2) { uint32_t g = __shfl(r, lane & ~7u); idx = (size_t)(g % nlines) * 8 + (lane & 7u); }
        else if (PATTERN == 3) { uint32_t g = __shfl(r, 0); idx = (size_t)(g % nlines) * 8; }
        else idx = (size_t)(r % nlines) * 8 + ((r >> 20 & 1) * 4);
        if (PATTERN == 4) {
            float4 a = table[idx], b = table[idx + 1], c = table[idx + 2], d = table[idx + 3];
            acc.x += a.x + b.y + c.z + d.w;
        } else {
            float4 a = table[idx];
            acc.x += a.x; acc.y += a.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int iters = 2000, blocks = 256 * 8, threads = 256;
    float4* out; hipMalloc(&out, sizeof(float4) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t kb : {16u, 2048u, 32768u}) {  // table size: L1-resident, L2-resident, MALL-resident
        uint32_t nlines = kb * 1024 / 128;
        float4* table; hipMalloc(&table, (size_t)nlines * 128); hipMemset(table, 0, (size_t)nlines * 128);
        for (int pat = 0; pat < 5; pat++) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                switch (pat) {
                    case 0: gather<0><<<blocks, threads>>>(table, nlines, iters, out, rep); break;
                    case 1: gather<1><<<blocks, threads>>>(table, nlines, iters, out, rep); break;
                    case 2: gather<2><<<blocks, threads>>>(table, nlines, iters, out, rep); break;
                    case 3: gather<3><<<blocks, threads>>>(table, nlines, iters, out, rep); break;
                    case 4: gather<4><<<blocks, threads>>>(table, nlines, iters, out, rep); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double lane_loads = (double)blocks * threads * iters * (pat == 4 ? 4 : 1);
            double per_cu_cycle = lane_loads / (best * 1e-3 * 2.4e9 * 256);
            printf("table %6u KB pattern %d: %.3f ms  %.2f lane-loads(16B)/clk/CU  %.1f GB/s/CU-equiv total %.2f TB/s\n", kb, pat, best,
                   per_cu_cycle, per_cu_cycle * 16 * 2.4, lane_loads * 16 / (best * 1e-3) / 1e12);
        }
        hipFree(table);
    }
    return 0;
}
