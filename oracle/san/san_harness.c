/*
 * san_harness.c — the sanitizer job of the CPU oracle (TEST INFRASTRUCTURE ONLY, like everything under oracle/).
 *
 * Drives every entry point of lbvh_oracle.c — the serial stages, their OpenMP forms (orc_*_mt, the threaded Morton /
 * tree / trace loops) and the path-tracing extension — on small seeded scenes whose sizes sit on the chunking borders,
 * with every buffer allocated at its exact size, and cross-checks serial against threaded results.  Compiled TOGETHER
 * with lbvh_oracle.c under -fsanitize=address,undefined (gcc) and under -fsanitize=thread with an OpenMP runtime that
 * tells the sanitizer about its barriers (clang + libomp + Archer) by oracle/san/Makefile; run by tests/test_sanitizers.py
 * in the CPU suite.  Never built or run on the GPU box's device side: sanitizers are a CPU-only tool in this project.
 *
 * Exit code 0 = every cross-check equal (and, by running to the end, no sanitizer report: the runs use
 * halt_on_error / -fno-sanitize-recover).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../lbvh_oracle.h"

static uint64_t rng_state;
static uint64_t splitmix64(void)
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float uniform(float lo, float hi) { return lo + (hi - lo) * (float)((splitmix64() >> 40) * (1.0 / 16777216.0)); }

static void* exact(size_t bytes)        /* exactly-sized, so that one element too far is a report */
{
    void* p = malloc(bytes ? bytes : 1);
    if (!p) { fprintf(stderr, "out of memory\n"); exit(3); }
    memset(p, 0xA5, bytes);
    return p;
}

#define CHECK(cond, what) do { if (!(cond)) { fprintf(stderr, "MISMATCH: %s (n = %u, threads = %d)\n", what, n, threads); return 1; } } while (0)

static int one_scene(uint32_t n, uint32_t capacity, int threads, int duplicates, int w, int h)
{
    const float mn[3] = {-125.0f, -125.0f, -125.0f}, mx[3] = {125.0f, 125.0f, 125.0f};
    lbvh_triangle* tris = exact((size_t)n * sizeof *tris);
    memset(tris, 0, (size_t)n * sizeof *tris);
    for (uint32_t i = 0; i < n; i++) {
        if (duplicates && i > 0 && (i % 3) != 0) { tris[i] = tris[i - 1]; continue; }      /* equal Morton codes */
        for (int k = 0; k < 3; k++) {
            const float c = uniform(-100.0f, 100.0f);
            tris[i].a[k] = c; tris[i].b[k] = c + uniform(-2.0f, 2.0f); tris[i].c[k] = c + uniform(-2.0f, 2.0f);
        }
        tris[i].b_uv[0] = 1.0f; tris[i].c_uv[1] = 1.0f;
        tris[i].a_normal[1] = tris[i].b_normal[1] = tris[i].c_normal[1] = 1.0f;
    }
    /* a-1 serial and threaded */
    uint32_t *keys = exact((size_t)capacity * 4), *idx = exact((size_t)capacity * 4), *keys2 = exact((size_t)capacity * 4), *idx2 = exact((size_t)capacity * 4);
    lbvh_aabb *aabb = exact((size_t)capacity * sizeof *aabb), *aabb2 = exact((size_t)capacity * sizeof *aabb2);
    orc_morton_aabb(tris, n, capacity, mn, mx, keys, idx, aabb, 1);
    orc_morton_aabb(tris, n, capacity, mn, mx, keys2, idx2, aabb2, threads);
    CHECK(memcmp(keys, keys2, (size_t)capacity * 4) == 0 && memcmp(idx, idx2, (size_t)capacity * 4) == 0, "morton serial vs threaded");
    CHECK(memcmp(aabb, aabb2, (size_t)n * sizeof *aabb) == 0, "aabb serial vs threaded");
    /* a-2..5 */
    orc_sort_pairs(keys, idx, capacity);
    orc_sort_pairs_mt(keys2, idx2, capacity, threads);
    CHECK(memcmp(keys, keys2, (size_t)capacity * 4) == 0 && memcmp(idx, idx2, (size_t)capacity * 4) == 0, "sort serial vs threaded");
    /* a-6 */
    orc_distribute_keys(keys, n);
    orc_distribute_keys_mt(keys2, n, threads);
    CHECK(memcmp(keys, keys2, (size_t)capacity * 4) == 0, "distribute serial vs threaded");
    /* a-7 */
    lbvh_internal_node *in1 = exact((size_t)capacity * sizeof *in1), *in2 = exact((size_t)capacity * sizeof *in2);
    lbvh_leaf_node *lf1 = exact((size_t)capacity * sizeof *lf1), *lf2 = exact((size_t)capacity * sizeof *lf2);
    memset(in1, 0xFF, (size_t)capacity * sizeof *in1); memset(in2, 0xFF, (size_t)capacity * sizeof *in2);
    memset(lf1, 0xFF, (size_t)capacity * sizeof *lf1); memset(lf2, 0xFF, (size_t)capacity * sizeof *lf2);
    CHECK(orc_build_tree(n, keys, in1, lf1, 1) == 0 && orc_build_tree(n, keys, in2, lf2, threads) == 0, "tree status");
    CHECK(memcmp(in1, in2, (size_t)capacity * sizeof *in1) == 0 && memcmp(lf1, lf2, (size_t)capacity * sizeof *lf1) == 0, "tree serial vs threaded");
    /* a-8 */
    lbvh_aabb *bvh1 = exact((size_t)capacity * sizeof *bvh1), *bvh2 = exact((size_t)capacity * sizeof *bvh2);
    CHECK(orc_refit(n, in1, lf1, aabb, idx, bvh1) == 0 && orc_refit_mt(n, in1, lf1, aabb, idx, bvh2, threads) == 0, "refit status");
    CHECK(memcmp(bvh1, bvh2, (size_t)(n - 1) * sizeof *bvh1) == 0, "refit serial vs threaded");
    /* the whole chain in one call */
    {
        uint32_t *k3 = exact((size_t)capacity * 4), *i3 = exact((size_t)capacity * 4);
        lbvh_aabb *a3 = exact((size_t)capacity * sizeof *a3), *b3 = exact((size_t)capacity * sizeof *b3);
        lbvh_internal_node* n3 = exact((size_t)capacity * sizeof *n3);
        lbvh_leaf_node* l3 = exact((size_t)capacity * sizeof *l3);
        memset(n3, 0xFF, (size_t)capacity * sizeof *n3); memset(l3, 0xFF, (size_t)capacity * sizeof *l3);
        CHECK(orc_build_all(tris, n, capacity, mn, mx, k3, i3, a3, n3, l3, b3, threads) == 0, "build_all status");
        CHECK(memcmp(k3, keys, (size_t)capacity * 4) == 0 && memcmp(n3, in1, (size_t)capacity * sizeof *n3) == 0 &&
              memcmp(b3, bvh1, (size_t)(n - 1) * sizeof *b3) == 0, "build_all vs stages");
        free(k3); free(i3); free(a3); free(b3); free(n3); free(l3);
    }
    /* a-9: whole frame, sub-rectangle with steps, serial vs threaded */
    lbvh_camera cam;
    cam.screen_width = w; cam.screen_height = h;
    cam.camera_fov = 0.57735026f; cam.near_plane = 0.3f;
    const float m[16] = {-1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 300, 0, 0, 0, 1};
    memcpy(cam.camera_to_world, m, sizeof m);
    lbvh_scene sc;
    sc.n = n; sc.sorted_indices = idx; sc.triangle_aabb = aabb; sc.internal_nodes = in1; sc.leaf_nodes = lf1; sc.bvh = bvh1; sc.triangles = tris;
    lbvh_hit *h1 = exact((size_t)w * h * sizeof *h1), *h2 = exact((size_t)w * h * sizeof *h2);
    lbvh_trace_stats s1, s2;
    CHECK(orc_trace_primary(&cam, 0, 0, w, h, 1, 1, &sc, h1, &s1, 1) == 0 && orc_trace_primary(&cam, 0, 0, w, h, 1, 1, &sc, h2, &s2, threads) == 0, "trace status");
    CHECK(memcmp(h1, h2, (size_t)w * h * sizeof *h1) == 0 && memcmp(&s1, &s2, sizeof s1) == 0, "trace serial vs threaded");
    {
        const int sw = (w - 3 + 2) / 3, sh = (h - 1 + 1) / 2;                 /* pixels x = 3, 6, ... ; y = 1, 3, ... */
        lbvh_hit* hs = exact((size_t)sw * sh * sizeof *hs);
        CHECK(orc_trace_primary(&cam, 3, 1, w, h, 3, 2, &sc, hs, NULL, threads) == 0, "strided trace status");
        for (int y = 0; y < sh; y++)
            for (int x = 0; x < sw; x++)
                CHECK(memcmp(&hs[y * sw + x], &h1[(1 + 2 * y) * w + 3 + 3 * x], sizeof *hs) == 0, "strided trace vs frame");
        free(hs);
    }
    /* shading tail */
    {
        uint8_t tex[4 * 4 * 4];
        for (int i = 0; i < 64; i++) tex[i] = (uint8_t)(i * 4);
        uint16_t* img = exact((size_t)w * h * 8);
        orc_shade(h1, (size_t)w * h, tris, tex, 4, 4, img);
        free(img);
    }
    /* extension: animate, path begin / trace / scatter / resolve */
    {
        uint32_t* body = exact((size_t)n * 4);
        for (uint32_t i = 0; i < n; i++) body[i] = i % 3;
        float centres[12] = {0};
        centres[0] = 40.0f; centres[5] = -40.0f; centres[10] = 40.0f;
        lbvh_triangle* moved = exact((size_t)n * sizeof *moved);
        orc_animate(tris, n, body, centres, 0.99500417f, 0.09983342f, moved);
        lbvh_path_state* st = exact((size_t)w * h * sizeof *st);
        orc_path_begin(&cam, st);
        lbvh_hit *r1 = exact((size_t)w * h * sizeof *r1), *r2 = exact((size_t)w * h * sizeof *r2);
        CHECK(orc_trace_rays(st, (size_t)w * h, 0.0f, &sc, r1, 1) == 0 && orc_trace_rays(st, (size_t)w * h, 0.0f, &sc, r2, threads) == 0, "trace_rays status");
        CHECK(memcmp(r1, r2, (size_t)w * h * sizeof *r1) == 0, "trace_rays serial vs threaded");
        for (uint32_t bounce = 0; bounce < 2; bounce++) {
            orc_path_scatter(&sc, r1, (size_t)w * h, bounce, 7u, 0.7f, st);
            CHECK(orc_trace_rays(st, (size_t)w * h, 1e-3f, &sc, r1, threads) == 0, "bounce status");
        }
        uint16_t* img = exact((size_t)w * h * 8);
        orc_path_resolve(st, (size_t)w * h, img);
        free(img); free(r1); free(r2); free(st); free(moved); free(body);
    }
    free(h1); free(h2); free(bvh1); free(bvh2); free(in1); free(in2); free(lf1); free(lf2);
    free(keys); free(idx); free(keys2); free(idx2); free(aabb); free(aabb2); free(tris);
    return 0;
}

int main(int argc, char** argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 4;
    const int big = argc > 2 ? atoi(argv[2]) : 20000;
    rng_state = 11;
    /* sizes on the borders: the smallest tree, one past a tile, a ragged capacity, more threads than elements */
    const uint32_t sizes[] = {2u, 3u, 7u, 1023u, 1024u, 1025u, (uint32_t)big};
    int bad = 0;
    for (size_t i = 0; i < sizeof sizes / sizeof sizes[0]; i++) {
        const uint32_t n = sizes[i], cap = (n + 1023u) / 1024u * 1024u;
        bad |= one_scene(n, cap, threads, 0, 33, 17);
        bad |= one_scene(n, cap, threads, 1, 16, 9);              /* duplicated triangles: equal Morton codes, deep trees */
    }
    /* the literal 32-lane emulation of the reference's five sort kernels at its smallest legal size */
    {
        const uint32_t tiles = 128, count = tiles * 1024;
        uint32_t *k = exact((size_t)count * 4), *v = exact((size_t)count * 4), *k2 = exact((size_t)count * 4), *v2 = exact((size_t)count * 4);
        for (uint32_t i = 0; i < count; i++) { k[i] = k2[i] = (uint32_t)splitmix64() >> (i % 5 == 0 ? 20 : 0); v[i] = v2[i] = i; }
        const int rc = orc_sort_pairs_literal(k, v, tiles);
        orc_sort_pairs(k2, v2, count);
        if (rc != 0 || memcmp(k, k2, (size_t)count * 4) != 0 || memcmp(v, v2, (size_t)count * 4) != 0) { fprintf(stderr, "MISMATCH: literal sort\n"); bad = 1; }
        free(k); free(v); free(k2); free(v2);
    }
    {
        const float bmin[3] = {-1, -1, -1}, bmax[3] = {1, 1, 1}, o[3] = {0, 0, 5}, inv[3] = {INFINITY, INFINITY, -1.0f};
        if (orc_ray_box(bmin, bmax, o, inv) != 1) { fprintf(stderr, "MISMATCH: ray_box\n"); bad = 1; }
    }
    printf("san_harness: %s (threads %d, largest scene %d triangles, OpenMP threads available %d)\n", bad ? "MISMATCH" : "ok", threads, big,
           orc_num_threads());
    return bad;
}
