// Microbenchmark: ways of getting one 128-byte node record per lane (random record per lane, as
// for incoherent secondary rays) from a table into the lane's registers on gfx950.
//   V0 lane7      every lane issues 7 x 16-byte loads of its own record (7 cache lookups per lane)
//   V1 coop_reg   8 loads per wave; in load k the 8 lanes of a group fetch the 8 x 16 B of the record
//                 of the group's k-th lane (one full line per 8 lanes); data left where it lands
//                 (vector-memory side only, not usable as is)
//   V2 coop_glds  as V1 but straight into LDS (global_load_lds_dwordx4), then every lane reads its
//                 own record back with 7 x ds_read_b128; parts are XOR-swizzled on the source side
//                 so that the read-back is bank-conflict-free
//   V3 coop_st    as V1 through registers, then ds_write_b128 + 7 x ds_read_b128
// `active` in (0,1]: fraction of lanes that need a record in an iteration (others idle).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/node_fetch.hip -o build/node_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int V>
__global__ __launch_bounds__(256) void fetch(const float4 *__restrict__ table, uint32_t nrec, int iters,
                                             uint32_t active_thresh, float *out) {
    __shared__ float4 stage[4][8 * 64];  // 8 KB per wave
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    float4 *my_stage = stage[wave];
    const uint32_t g8 = lane & ~7u, p = lane & 7u, k_me = lane & 7u, g = lane >> 3;
    for (int i = 0; i < iters; i++) {
        const uint32_t r = lcg(s);
        const bool act = (lcg(s) & 0xFFFFu) < active_thresh;
        const uint32_t rec = r % nrec;
        if (V == 0) {
            if (act) {
                const float4 *np = table + (size_t)rec * 8;
                float4 a = np[0], b = np[1], c = np[2], d = np[3], e = np[4], f = np[5], h = np[6];
                acc += a.x + b.y + c.z + d.w + e.x + f.y + h.z;
            }
        } else {
            // record index and activity of the group's 8 lanes
            uint32_t code = act ? rec : 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t ck = __shfl(code, g8 + k);
                if (ck != 0xFFFFFFFFu) {
                    const uint32_t part = (V == 1) ? p : (p ^ (uint32_t)k);
                    const float4 *src = table + (size_t)ck * 8 + part;
                    if (V == 2) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                         (__attribute__((address_space(3))) void *)(my_stage + k * 64), 16, 0, 0);
                    } else if (V == 1) {
                        float4 a = *src;
                        acc += a.x;
                    } else {
                        float4 a = *src;
                        my_stage[k * 64 + lane] = a;
                    }
                }
            }
            if (V >= 2) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (act) {
                    const float4 *mine = my_stage + k_me * 64 + g * 8;
                    float4 a = mine[0 ^ k_me], b = mine[1 ^ k_me], c = mine[2 ^ k_me], d = mine[3 ^ k_me],
                           e = mine[4 ^ k_me], f = mine[5 ^ k_me], h = mine[6 ^ k_me];
                    acc += a.x + b.y + c.z + d.w + e.x + f.y + h.z;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int iters = 400, blocks = 256 * 5, threads = 256;
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * threads));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (uint32_t kb : {16u, 2048u, 17408u}) {  // L1-resident, L2-resident, L2+Infinity-Cache (size of the bench scene's nodes)
        const uint32_t nrec = kb * 1024 / 128;
        float4 *table;
        CHECK(hipMalloc(&table, (size_t)nrec * 128));
        std::vector<float> h((size_t)nrec * 32);
        for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 977) * 0.001f;
        CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (float active : {1.0f, 0.6f}) {
            const uint32_t th = (uint32_t)(active * 65536.0f);
            double sums[4];
            for (int v = 0; v < 4; v++) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; rep++) {
                    CHECK(hipEventRecord(e0));
                    switch (v) {
                        case 0: fetch<0><<<blocks, threads>>>(table, nrec, iters, th, out); break;
                        case 1: fetch<1><<<blocks, threads>>>(table, nrec, iters, th, out); break;
                        case 2: fetch<2><<<blocks, threads>>>(table, nrec, iters, th, out); break;
                        case 3: fetch<3><<<blocks, threads>>>(table, nrec, iters, th, out); break;
                    }
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                std::vector<float> ho((size_t)blocks * threads);
                CHECK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
                double sum = 0;
                for (float x : ho) sum += x;
                sums[v] = sum;
                const double recs = (double)blocks * threads * iters * active;
                printf("table %6u KB active %.1f V%d: %8.3f ms  %.3f records/clk/CU (2.4 GHz)  %.2f G records/s\n", kb, active,
                       v, best, recs / (best * 1e-3 * 2.4e9 * 256), recs / (best * 1e-3) / 1e9);
            }
            printf("   checksums V0 %.6g V2 %.6g V3 %.6g %s\n", sums[0], sums[2], sums[3],
                   (sums[0] == sums[2] && sums[0] == sums[3]) ? "(equal)" : "(DIFFER)");
        }
        CHECK(hipFree(table));
    }
    return 0;
}
