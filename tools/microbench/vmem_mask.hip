// Microbenchmark: what a 16-byte-per-lane (and a 4-byte-per-lane) global load costs the CU's
// vector-memory path as a function of WHICH lanes are active.  All lanes read from a small
// L1-resident table (one 128-byte line per group of 4 lanes), so neither tag lookups nor misses
// limit; what remains is address processing and data return.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/vmem_mask.hip -o build/vmem_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int WIDTH>
__global__ __launch_bounds__(256) void loads(const float4 *__restrict__ table, int iters, unsigned long long mask, float *out) {
    const uint32_t lane = threadIdx.x & 63u;
    float acc = 0.f;
    if ((mask >> lane) & 1ull) {
        const float4 *p = table + (lane >> 2) * 8;  // quad q reads line q, all of it
        for (int i = 0; i < iters; i++) {
            // written as inline assembly so that the compiler neither merges nor hoists the loads
            typedef float vec4f __attribute__((ext_vector_type(4)));
            if (WIDTH == 4) {
                vec4f a, b, c, d;
                asm volatile(
                    "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                    "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\t"
                    "global_load_dwordx4 %0, %4, off offset:64\n\tglobal_load_dwordx4 %1, %4, off offset:80\n\t"
                    "global_load_dwordx4 %2, %4, off offset:96\n\tglobal_load_dwordx4 %3, %4, off offset:112\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                    : "v"(p)
                    : "memory");
                acc += a.x + b.x + c.x + d.x;
            } else if (WIDTH == 3) {
                typedef float vec3f __attribute__((ext_vector_type(3)));
                vec3f a, b, c, d;
                asm volatile(
                    "global_load_dwordx3 %0, %4, off\n\tglobal_load_dwordx3 %1, %4, off offset:12\n\t"
                    "global_load_dwordx3 %2, %4, off offset:24\n\tglobal_load_dwordx3 %3, %4, off offset:36\n\t"
                    "global_load_dwordx3 %0, %4, off offset:48\n\tglobal_load_dwordx3 %1, %4, off offset:60\n\t"
                    "global_load_dwordx3 %2, %4, off offset:72\n\tglobal_load_dwordx3 %3, %4, off offset:84\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                    : "v"(p)
                    : "memory");
                acc += a.x + b.x + c.x + d.x;
            } else if (WIDTH == 2) {
                typedef float vec2f __attribute__((ext_vector_type(2)));
                vec2f a, b, c, d;
                asm volatile(
                    "global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %4, off offset:16\n\t"
                    "global_load_dwordx2 %2, %4, off offset:32\n\tglobal_load_dwordx2 %3, %4, off offset:48\n\t"
                    "global_load_dwordx2 %0, %4, off offset:64\n\tglobal_load_dwordx2 %1, %4, off offset:80\n\t"
                    "global_load_dwordx2 %2, %4, off offset:96\n\tglobal_load_dwordx2 %3, %4, off offset:112\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                    : "v"(p)
                    : "memory");
                acc += a.x + b.x + c.x + d.x;
            } else {
                float a, b, c, d;
                asm volatile(
                    "global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:16\n\t"
                    "global_load_dword %2, %4, off offset:32\n\tglobal_load_dword %3, %4, off offset:48\n\t"
                    "global_load_dword %0, %4, off offset:64\n\tglobal_load_dword %1, %4, off offset:80\n\t"
                    "global_load_dword %2, %4, off offset:96\n\tglobal_load_dword %3, %4, off offset:112\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                    : "v"(p)
                    : "memory");
                acc += a + b + c + d;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int iters = 2000, blocks = 256 * 8, threads = 256;
    float *out;
    float4 *table;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * threads));
    CHECK(hipMalloc(&table, 16 * 128 + 256));
    CHECK(hipMemset(table, 0, 16 * 128 + 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct { const char *name; unsigned long long mask; } pats[] = {
        {"64 lanes", ~0ull},
        {"32 lanes, contiguous (8 quads)", 0xFFFFFFFFull},
        {"32 lanes, every other (16 quads)", 0x5555555555555555ull},
        {"16 lanes, contiguous (4 quads)", 0xFFFFull},
        {"16 lanes, one per quad (16 quads)", 0x1111111111111111ull},
        {"8 lanes, contiguous (2 quads)", 0xFFull},
        {"8 lanes, one per other quad (8 quads)", 0x0101010101010101ull},
        {"4 lanes, contiguous (1 quad)", 0xFull},
        {"4 lanes, spread (4 quads)", 0x0001000100010001ull},
        {"1 lane", 0x1ull},
        {"6 lanes, spread (6 quads)", 0x0000010010010111ull},
    };
    for (int width : {4, 3, 2, 1}) {
        for (auto &pt : pats) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                if (width == 4) loads<4><<<blocks, threads>>>(table, iters, pt.mask, out);
                else if (width == 3) loads<3><<<blocks, threads>>>(table, iters, pt.mask, out);
                else if (width == 2) loads<2><<<blocks, threads>>>(table, iters, pt.mask, out);
                else loads<1><<<blocks, threads>>>(table, iters, pt.mask, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double instr = (double)blocks * (threads / 64) * iters * 8;
            printf("%2d B/lane  %-40s %8.3f ms  %6.2f clk per wave-instruction per CU (2.4 GHz)\n", width * 4, pt.name, best,
                   best * 1e-3 * 2.4e9 * 256 / instr);
        }
    }
    return 0;
}
