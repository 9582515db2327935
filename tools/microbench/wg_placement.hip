// Where does the dispatcher put the workgroups of a launch shaped like the trace kernel's (256
// threads, 20 KB LDS, 8160 workgroups)?  Every workgroup records the XCD, shader engine and CU it
// runs on and when it started; the host prints the placement of the first workgroups and how often
// workgroups that are consecutive within their XCD (b, b + 8, b + 16, ...) share a CU.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/wg_placement.hip -o build/wg_placement && build/wg_placement
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec { uint32_t hw_id, xcc_id; unsigned long long t0, t1; };

__global__ __launch_bounds__(256) void probe(Rec *out, int spin) {
    __shared__ float pad[5120];  // 20 KB, as the trace kernel
    pad[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    float a = pad[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; i++) a = a * 1.0000001f + 0.5f;
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        Rec r;
        r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        r.t0 = t0, r.t1 = t1;
        out[blockIdx.x] = r;
    }
    if (a == 12345.678f) out[0].hw_id = 0;  // keeps the loop
}

int main() {
    const int n = 8160;
    Rec *d;
    CHECK(hipMalloc(&d, sizeof(Rec) * n));
    std::vector<Rec> h(n);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d, 20000);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), d, sizeof(Rec) * n, hipMemcpyDeviceToHost));
    auto cu_of = [](const Rec &r) { return (r.hw_id >> 8) & 0xF; };
    auto sh_of = [](const Rec &r) { return (r.hw_id >> 12) & 0x1; };
    auto se_of = [](const Rec &r) { return (r.hw_id >> 13) & 0x7; };
    auto xcc_of = [](const Rec &r) { return r.xcc_id & 0xF; };
    auto where = [&](const Rec &r) { return (xcc_of(r) << 12) | (se_of(r) << 8) | (sh_of(r) << 4) | cu_of(r); };
    printf("first 48 workgroups: blockIdx -> xcc.se.sh.cu\n");
    for (int b = 0; b < 48; b++) printf("%d->%u.%u.%u.%u%s", b, xcc_of(h[b]), se_of(h[b]), sh_of(h[b]), cu_of(h[b]), (b % 8 == 7) ? "\n" : "  ");
    printf("workgroups 0, 8, 16, ... (consecutive within XCD of block 0): se.sh.cu\n");
    for (int j = 0; j < 72; j++) { const Rec &r = h[j * 8]; printf("%u.%u.%u%s", se_of(r), sh_of(r), cu_of(r), (j % 12 == 11) ? "\n" : " "); }
    // does b & 7 name the XCD?
    int same = 0;
    for (int b = 0; b < n; b++) same += xcc_of(h[b]) == xcc_of(h[b & 7]);
    printf("blocks whose XCD equals that of block (b & 7): %d of %d\n", same, n);
    // how often do the four workgroups of a 2x2 macro tile (j = 4k .. 4k+3 within an XCD) share a CU?
    int groups = 0, all_same = 0, pairs_same = 0, pairs = 0;
    for (int x = 0; x < 8; x++)
        for (int k = 0; (k * 4 + 3) * 8 + x < n; k++) {
            uint32_t w[4];
            for (int i = 0; i < 4; i++) w[i] = where(h[(k * 4 + i) * 8 + x]);
            groups++;
            all_same += (w[0] == w[1] && w[1] == w[2] && w[2] == w[3]);
            for (int i = 0; i < 4; i++) for (int m = i + 1; m < 4; m++) { pairs++; pairs_same += w[i] == w[m]; }
        }
    printf("macro-tile groups of 4 consecutive workgroups: %d, all on one CU: %d, pairs on one CU: %d of %d\n", groups, all_same, pairs_same, pairs);
    std::map<uint32_t, int> per_cu;
    for (int b = 0; b < n; b++) per_cu[where(h[b])]++;
    int mn = 1 << 30, mx = 0;
    for (auto &kv : per_cu) { mn = kv.second < mn ? kv.second : mn; mx = kv.second > mx ? kv.second : mx; }
    printf("distinct CUs seen: %zu, workgroups per CU min %d max %d\n", per_cu.size(), mn, mx);
    return 0;
}
