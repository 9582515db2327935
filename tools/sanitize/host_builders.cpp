// Address / undefined-behaviour sanitizer run of the host-side tree builders (scene_builder.cpp,
// rt_fastbvh.cpp) on random and degenerate inputs.  Host code only; no GPU involved.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all \
//       tools/sanitize/host_builders.cpp raytracer_amd/csrc/scene_builder.cpp raytracer_amd/csrc/rt_fastbvh.cpp \
//       -o build/sanitize_host_builders && build/sanitize_host_builders
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

extern "C" {
const char *rts_last_error(void);
int rts_build_qbvh(const float *aabb_min, const float *aabb_max, const uint8_t *types, const uint32_t *indices, uint32_t n,
                   uint32_t **nodes_out, uint32_t *n_nodes_out, float scene_min[3], float scene_extent[3]);
void rts_free(void *p);
int rts_build_fast_bvh(const uint32_t *tris, uint32_t n_tris, const uint32_t *spheres, uint32_t n_spheres,
                       const uint32_t *boxes, uint32_t n_boxes, uint32_t **nodes_out, uint32_t *n_nodes_out);
}

static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// kind 0: random triangles; 1: all primitives identical (every split degenerate); 2: all on one
// plane and one line; 3: huge coordinates mixed with tiny ones
static int run_case(uint32_t n, int kind, uint32_t seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-10.f, 10.f), S(0.001f, 0.5f);
    std::vector<uint32_t> tris((size_t)n * 16, 0u), spheres, boxes;
    std::vector<float> mn((size_t)n * 3), mx((size_t)n * 3);
    std::vector<uint8_t> types(n, 0);
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; i++) {
        float v[3][3];
        for (int a = 0; a < 3; a++) {
            float c = kind == 1 ? 1.0f : kind == 2 ? (a == 0 ? U(rng) : 2.0f) : U(rng);
            if (kind == 3 && (i & 3) == 0) c *= 1e6f;
            for (int k = 0; k < 3; k++) v[k][a] = c + (kind == 1 || kind == 2 ? 0.0f : S(rng));
        }
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) tris[(size_t)i * 16 + 4 * k + a] = f2u(v[k][a]);
        for (int a = 0; a < 3; a++) {
            mn[(size_t)i * 3 + a] = std::fmin(v[0][a], std::fmin(v[1][a], v[2][a]));
            mx[(size_t)i * 3 + a] = std::fmax(v[0][a], std::fmax(v[1][a], v[2][a]));
        }
        idx[i] = i;
    }
    uint32_t nsph = n / 7, nbox = n / 11;
    for (uint32_t i = 0; i < nsph; i++) {
        float r = S(rng);
        uint32_t w[8] = {f2u(U(rng)), f2u(U(rng)), f2u(U(rng)), f2u(r), f2u(r * r), 0u, 0u, 0u};
        spheres.insert(spheres.end(), w, w + 8);
    }
    for (uint32_t i = 0; i < nbox; i++) {
        float a = U(rng), b = U(rng), c = U(rng);
        uint32_t w[8] = {f2u(a), f2u(b), f2u(c), 0u, f2u(a + S(rng)), f2u(b + S(rng)), f2u(c + S(rng)), 0u};
        boxes.insert(boxes.end(), w, w + 8);
    }
    uint32_t *nodes = nullptr, nn = 0;
    float smin[3], sext[3];
    if (rts_build_qbvh(mn.data(), mx.data(), types.data(), idx.data(), n, &nodes, &nn, smin, sext)) {
        if (n == 0) return 0;  // refused by design
        printf("qbvh n=%u kind=%d: %s\n", n, kind, rts_last_error());
        return 1;
    }
    // every primitive appears exactly once among the leaf words
    std::vector<uint32_t> seen(n, 0);
    for (uint32_t i = 0; i < nn; i++)
        for (int k = 0; k < 4; k++) {
            uint32_t w = nodes[(size_t)i * 16 + 12 + k];
            if ((w >> 30) == 0u && w != 0xFFFFFFFFu) {
                if ((w & 0x3FFFFFFFu) >= n) { printf("leaf index out of range\n"); return 1; }
                seen[w & 0x3FFFFFFFu]++;
            } else if ((w >> 30) == 3u && w != 0xFFFFFFFFu && (w & 0x3FFFFFFFu) >= nn) {
                printf("child index out of range\n");
                return 1;
            }
        }
    if (n > 1)
        for (uint32_t i = 0; i < n; i++)
            if (seen[i] != 1) { printf("qbvh n=%u kind=%d: primitive %u appears %u times\n", n, kind, i, seen[i]); return 1; }
    rts_free(nodes);
    uint32_t *fn = nullptr, fnn = 0;
    if (rts_build_fast_bvh(tris.data(), n, spheres.data(), nsph, boxes.data(), nbox, &fn, &fnn)) {
        printf("fast n=%u kind=%d: %s\n", n, kind, rts_last_error());
        return 1;
    }
    rts_free(fn);
    return 0;
}

// Fast-mode builder only.  kind 0: centroids spaced geometrically along x (a SAH split peels off
// one primitive at a time: the recursion depth must stay bounded); kind 1: coordinates at the end
// of the float range (extents overflow: the builder must refuse the scene, not produce boxes that
// are not conservative).
static int run_fast_case(uint32_t n, int kind) {
    std::vector<uint32_t> tris((size_t)n * 16, 0u);
    for (uint32_t i = 0; i < n; i++) {
        const float x = kind == 0 ? std::ldexp(1.0f, -100 + (int)(i % 200u)) * (1.0f + (float)(i / 200u) * 1e-4f)
                                  : ((i & 1u) ? 3.0e38f : -3.0e38f);
        const float v[3][3] = {{x, 0.f, 0.f}, {x * 1.0001f, 1.f, 0.f}, {x, 0.f, 1.f}};
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) tris[(size_t)i * 16 + 4 * k + a] = f2u(v[k][a]);
    }
    uint32_t *fn = nullptr, fnn = 0;
    const int rc = rts_build_fast_bvh(tris.data(), n, nullptr, 0, nullptr, 0, &fn, &fnn);
    if (kind == 1) {
        if (rc == 0) { printf("fast: a scene whose extents overflow was accepted\n"); rts_free(fn); return 1; }
        return 0;
    }
    if (rc) { printf("fast skewed n=%u: %s\n", n, rts_last_error()); return 1; }
    rts_free(fn);
    return 0;
}

int main() {
    int bad = 0, cases = 0;
    bad += run_fast_case(200000u, 0), cases++;
    bad += run_fast_case(64u, 1), cases++;
    for (int kind = 0; kind < 4; kind++)
        for (uint32_t n : {0u, 1u, 2u, 3u, 4u, 5u, 17u, 64u, 1000u, 20000u}) {
            bad += run_case(n, kind, 1234u + n + 7u * (uint32_t)kind);
            cases++;
        }
    printf("%d cases, %d failed\n", cases, bad);
    return bad ? 1 : 0;
}
