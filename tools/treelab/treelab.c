// treelab.c — CPU laboratory for the DERIVED traversal tree: builds candidate trees over the Morton-sorted leaves of a
// scene and counts what the GPU walkers would do on them — packet steps of walk_packet_lean's control flow (8x8-pixel tiles,
// leaves first, majority-vote near child, per-ray t pruning) and node visits of a single-ray near-first walk (diffuse bounce rays,
// the cfg5 proxy).  A tool for choosing an algorithm BEFORE writing a kernel; not part of the product, not an oracle.
//   gcc -O2 -fopenmp -o treelab treelab.c -lm ;  ./treelab scene.tri [mode ...]
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LEAF 0x80000000u
#define MAXF 2139095040.0f

typedef struct { float mn[3], mx[3]; } box_t;
typedef struct { uint32_t l, r; box_t lb, rb; } node_t;
typedef struct { node_t* nd; uint32_t count; } tree_t;

static uint32_t n;
static float (*tri)[9];      // original order: a, b, c
static box_t* lbox;          // leaf boxes, sorted order
static uint32_t* sidx;       // sorted position -> original triangle
static uint32_t* akeys;      // aligned keys (strictly increasing)
static uint32_t* mkeys;      // sorted raw Morton codes

static inline box_t box_empty(void) { box_t b = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}}; return b; }
static inline box_t box_union(box_t a, box_t b)
{
    for (int k = 0; k < 3; k++) { a.mn[k] = fminf(a.mn[k], b.mn[k]); a.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
    return a;
}
static inline float box_area(box_t b)
{
    const float x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
    return x * y + y * z + z * x;
}
static inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

static uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static uint32_t morton(float x, float y, float z)
{
    x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f); y = fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f); z = fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    return expand_bits((uint32_t)x) * 4 + expand_bits((uint32_t)y) * 2 + expand_bits((uint32_t)z);
}

typedef struct { uint32_t key, idx; } pair_t;
static int cmp_pair(const void* a, const void* b)
{
    const pair_t *p = a, *q = b;
    if (p->key != q->key) return p->key < q->key ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx);
}

static void load_scene(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f || fread(&n, 4, 1, f) != 1) { fprintf(stderr, "cannot read %s\n", path); exit(1); }
    tri = malloc((size_t)n * 36);
    if (fread(tri, 36, n, f) != n) { fprintf(stderr, "short file\n"); exit(1); }
    fclose(f);
    box_t* ob = malloc((size_t)n * sizeof(box_t));
    pair_t* p = malloc((size_t)n * sizeof(pair_t));
    for (uint32_t i = 0; i < n; i++) {
        box_t b;
        for (int k = 0; k < 3; k++) {
            b.mn[k] = fminf(fminf(tri[i][k], tri[i][3 + k]), tri[i][6 + k]) - 0.001f;
            b.mx[k] = fmaxf(fmaxf(tri[i][k], tri[i][3 + k]), tri[i][6 + k]) + 0.001f;
        }
        ob[i] = b;
        float c[3];
        for (int k = 0; k < 3; k++) c[k] = ((b.mn[k] + b.mx[k]) * 0.5f - (-125.0f)) / 250.0f;
        p[i].key = morton(c[0], c[1], c[2]);
        p[i].idx = i;
    }
    qsort(p, n, sizeof(pair_t), cmp_pair);
    lbox = malloc((size_t)n * sizeof(box_t)); sidx = malloc((size_t)n * 4); akeys = malloc((size_t)n * 4); mkeys = malloc((size_t)n * 4);
    int64_t run = INT64_MIN;
    for (uint32_t i = 0; i < n; i++) {
        lbox[i] = ob[p[i].idx]; sidx[i] = p[i].idx; mkeys[i] = p[i].key;
        const int64_t v = (int64_t)p[i].key - (int64_t)i;
        if (v > run) run = v;
        akeys[i] = (uint32_t)(run + (int64_t)i);
    }
    free(ob); free(p);
}

// ---- builders over the sorted order (contiguous ranges) ----------------------------------------------------------------------
enum { SPLIT_RADIX, SPLIT_SWEEP, SPLIT_MEDIAN, SPLIT_RADIX_SAH };
static int g_top_mode = SPLIT_RADIX, g_bottom_mode = SPLIT_RADIX;
static uint32_t g_threshold = 0;          // ranges of more than this many leaves use the top mode
static uint32_t g_cand_levels = 3;        // SPLIT_RADIX_SAH: candidates = the radix splits of this many levels
static float* g_suffix;                   // scratch of the sweep
static const uint32_t* g_keys;

static uint32_t radix_split(uint32_t a, uint32_t b)
{
    const uint32_t fc = g_keys[a], lc = g_keys[b];
    if (fc == lc) return (a + b) >> 1;
    const int common = clz32(fc ^ lc);
    uint32_t lo = a, hi = b;              // last position in [a, b) whose key shares more than `common` bits with the first
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (clz32(fc ^ g_keys[mid]) > common) lo = mid; else hi = mid;
    }
    return lo;
}

static uint32_t sweep_split(uint32_t a, uint32_t b)
{
    box_t acc = box_empty();
    for (uint32_t k = b; k > a; k--) { acc = box_union(acc, lbox[k]); g_suffix[k] = box_area(acc); }
    acc = box_empty();
    float best = INFINITY; uint32_t at = a;
    for (uint32_t k = a; k < b; k++) {
        acc = box_union(acc, lbox[k]);
        const float c = box_area(acc) * (float)(k - a + 1) + g_suffix[k + 1] * (float)(b - k);
        if (c < best) { best = c; at = k; }
    }
    return at;
}

static box_t range_box(uint32_t a, uint32_t b)
{
    box_t acc = box_empty();
    for (uint32_t k = a; k <= b; k++) acc = box_union(acc, lbox[k]);
    return acc;
}

// candidates: the radix splits of the first g_cand_levels levels below [a, b]; the one with the least SAH cost
static void radix_candidates(uint32_t a, uint32_t b, uint32_t depth, uint32_t* out, uint32_t* count)
{
    if (a >= b) return;
    const uint32_t s = radix_split(a, b);
    out[(*count)++] = s;
    if (depth + 1 < g_cand_levels) { radix_candidates(a, s, depth + 1, out, count); radix_candidates(s + 1, b, depth + 1, out, count); }
}
static uint32_t radix_sah_split(uint32_t a, uint32_t b)
{
    uint32_t cand[256], count = 0;
    radix_candidates(a, b, 0, cand, &count);
    float best = INFINITY; uint32_t at = cand[0];
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t k = cand[i];
        const float c = box_area(range_box(a, k)) * (float)(k - a + 1) + box_area(range_box(k + 1, b)) * (float)(b - k);
        if (c < best) { best = c; at = k; }
    }
    return at;
}

static uint32_t build_range(tree_t* t, uint32_t a, uint32_t b, box_t* out)
{
    if (a == b) { *out = lbox[a]; return LEAF | a; }
    const uint32_t idx = t->count++;
    const int mode = (b - a + 1) > g_threshold ? g_top_mode : g_bottom_mode;
    uint32_t s;
    switch (mode) {
    case SPLIT_SWEEP: s = sweep_split(a, b); break;
    case SPLIT_MEDIAN: s = (a + b) >> 1; break;
    case SPLIT_RADIX_SAH: s = radix_sah_split(a, b); break;
    default: s = radix_split(a, b);
    }
    box_t lb, rb;
    const uint32_t l = build_range(t, a, s, &lb);
    const uint32_t r = build_range(t, s + 1, b, &rb);
    t->nd[idx].l = l; t->nd[idx].r = r; t->nd[idx].lb = lb; t->nd[idx].rb = rb;
    *out = box_union(lb, rb);
    return idx;
}

// ---- binned SAH over an arbitrary permutation of the leaves (the quality ceiling) --------------------------------------------
static uint32_t* g_perm;
#define BINS 32
static uint32_t build_binned(tree_t* t, uint32_t a, uint32_t b, box_t* out)     // leaves g_perm[a..b]
{
    if (a == b) { *out = lbox[g_perm[a]]; return LEAF | g_perm[a]; }
    const uint32_t idx = t->count++;
    box_t cb = box_empty();
    for (uint32_t k = a; k <= b; k++)
        for (int d = 0; d < 3; d++) {
            const float c = (lbox[g_perm[k]].mn[d] + lbox[g_perm[k]].mx[d]) * 0.5f;
            cb.mn[d] = fminf(cb.mn[d], c); cb.mx[d] = fmaxf(cb.mx[d], c);
        }
    float best = INFINITY; int best_axis = -1, best_bin = 0;
    for (int d = 0; d < 3; d++) {
        const float ext = cb.mx[d] - cb.mn[d];
        if (!(ext > 0.0f)) continue;
        box_t bb[BINS]; uint32_t cnt[BINS];
        for (int i = 0; i < BINS; i++) { bb[i] = box_empty(); cnt[i] = 0; }
        const float scale = (float)BINS / ext;
        for (uint32_t k = a; k <= b; k++) {
            const box_t* q = &lbox[g_perm[k]];
            int bi = (int)(((q->mn[d] + q->mx[d]) * 0.5f - cb.mn[d]) * scale);
            if (bi >= BINS) bi = BINS - 1;
            if (bi < 0) bi = 0;
            bb[bi] = box_union(bb[bi], *q); cnt[bi]++;
        }
        float ra[BINS]; uint32_t rc[BINS];
        box_t acc = box_empty(); uint32_t c = 0;
        for (int i = BINS - 1; i > 0; i--) { acc = box_union(acc, bb[i]); c += cnt[i]; ra[i] = box_area(acc); rc[i] = c; }
        acc = box_empty(); c = 0;
        for (int i = 0; i < BINS - 1; i++) {
            acc = box_union(acc, bb[i]); c += cnt[i];
            if (c == 0 || rc[i + 1] == 0) continue;
            const float cost = box_area(acc) * (float)c + ra[i + 1] * (float)rc[i + 1];
            if (cost < best) { best = cost; best_axis = d; best_bin = i; }
        }
    }
    uint32_t mid;
    if (best_axis < 0) {
        mid = (a + b) >> 1;
    } else {
        const int d = best_axis;
        const float scale = (float)BINS / (cb.mx[d] - cb.mn[d]);
        uint32_t i = a, j = b;
        for (;;) {
            while (i <= j) {
                int bi = (int)(((lbox[g_perm[i]].mn[d] + lbox[g_perm[i]].mx[d]) * 0.5f - cb.mn[d]) * scale);
                if (bi >= BINS) bi = BINS - 1;
                if (bi > best_bin) break;
                i++;
            }
            while (i < j) {
                int bi = (int)(((lbox[g_perm[j]].mn[d] + lbox[g_perm[j]].mx[d]) * 0.5f - cb.mn[d]) * scale);
                if (bi >= BINS) bi = BINS - 1;
                if (bi <= best_bin) break;
                j--;
            }
            if (i >= j) break;
            const uint32_t tmp = g_perm[i]; g_perm[i] = g_perm[j]; g_perm[j] = tmp;
        }
        mid = i - 1;                   // leaves a..mid left
        if (i == a || i > b) mid = (a + b) >> 1;
    }
    box_t lb, rb;
    const uint32_t l = build_binned(t, a, mid, &lb);
    const uint32_t r = build_binned(t, mid + 1, b, &rb);
    t->nd[idx].l = l; t->nd[idx].r = r; t->nd[idx].lb = lb; t->nd[idx].rb = rb;
    *out = box_union(lb, rb);
    return idx;
}


// ---- PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) over the Morton order: the bottom-up ceiling --------------
typedef struct { uint32_t ref; box_t b; } cluster_t;
static void build_ploc(tree_t* t, uint32_t radius)
{
    cluster_t* c = malloc((size_t)n * sizeof(cluster_t));
    cluster_t* c2 = malloc((size_t)n * sizeof(cluster_t));
    uint32_t* nn = malloc((size_t)n * 4);
    uint32_t m = n;
    for (uint32_t i = 0; i < n; i++) { c[i].ref = LEAF | i; c[i].b = lbox[i]; }
    // nodes are created bottom-up: allocate from the END of the array so that the root, created last, is node 0
    uint32_t next = n - 1;
    while (m > 1) {
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < m; i++) {
            const uint32_t lo = i > radius ? i - radius : 0, hi = i + radius < m - 1 ? i + radius : m - 1;
            float best = INFINITY; uint32_t at = i;
            for (uint32_t j = lo; j <= hi; j++) {
                if (j == i) continue;
                const float a = box_area(box_union(c[i].b, c[j].b));
                if (a < best) { best = a; at = j; }
            }
            nn[i] = at;
        }
        uint32_t out = 0;
        for (uint32_t i = 0; i < m; i++) {
            const uint32_t j = nn[i];
            if (nn[j] == i) {
                if (i < j) {
                    const uint32_t idx = --next;
                    t->nd[idx].l = c[i].ref; t->nd[idx].r = c[j].ref; t->nd[idx].lb = c[i].b; t->nd[idx].rb = c[j].b;
                    c2[out].ref = idx; c2[out].b = box_union(c[i].b, c[j].b); out++;
                }
            } else c2[out++] = c[i];
        }
        cluster_t* tmp = c; c = c2; c2 = tmp;
        m = out;
    }
    t->count = n - 1;
    free(c); free(c2); free(nn);
}

// ---- tree statistics ---------------------------------------------------------------------------------------------------------
static double tree_sah(const tree_t* t, uint32_t* depth_out)
{
    // sum of the areas of all internal nodes' child boxes over the root's area (the leaf term is the same for every topology)
    double sum = 0.0;
    box_t root = box_union(t->nd[0].lb, t->nd[0].rb);
    for (uint32_t i = 0; i < t->count; i++) {
        if (!(t->nd[i].l & LEAF)) sum += box_area(t->nd[i].lb);
        if (!(t->nd[i].r & LEAF)) sum += box_area(t->nd[i].rb);
    }
    // depth
    uint32_t maxd = 0;
    uint32_t* st = malloc(sizeof(uint32_t) * 2 * 4096);
    uint32_t sp = 0; st[0] = 0; st[1] = 1; sp = 1;
    while (sp) {
        sp--;
        const uint32_t i = st[2 * sp], d = st[2 * sp + 1];
        if (d > maxd) maxd = d;
        if (!(t->nd[i].l & LEAF)) { st[2 * sp] = t->nd[i].l; st[2 * sp + 1] = d + 1; sp++; }
        if (!(t->nd[i].r & LEAF)) { st[2 * sp] = t->nd[i].r; st[2 * sp + 1] = d + 1; sp++; }
    }
    free(st);
    *depth_out = maxd;
    return sum / box_area(root);
}

// ---- rays --------------------------------------------------------------------------------------------------------------------
typedef struct { float o[3], d[3], inv[3]; } ray_t;
typedef struct { int w, h; float fov, near_plane; float pos[3]; } camera_t;

static ray_t camera_ray(const camera_t* c, uint32_t px, uint32_t py)
{
    const float height = 2.0f * c->near_plane * c->fov, width = (float)c->w * height / (float)c->h;
    const float dx = -width / 2.0f + width / (float)c->w * ((float)px + 0.5f);
    const float dy = -height / 2.0f + height / (float)c->h * ((float)py + 0.5f);
    const float dz = -c->near_plane;
    // cameraToWorld rows [-1,0,0,px], [0,1,0,py], [0,0,1,pz]
    float w[3] = {-dx, dy, dz};
    const float len = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    ray_t r;
    for (int k = 0; k < 3; k++) { r.o[k] = c->pos[k]; r.d[k] = w[k] / len; r.inv[k] = 1.0f / r.d[k]; }
    return r;
}

static inline int slab(const box_t* b, const ray_t* r, float* tmin_out)
{
    float tmin = -INFINITY, tmax = INFINITY;
    for (int k = 0; k < 3; k++) {
        const float t1 = (b->mn[k] - r->o[k]) * r->inv[k], t2 = (b->mx[k] - r->o[k]) * r->inv[k];
        tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    }
    *tmin_out = tmin;
    return tmax > fmaxf(tmin, 0.0f);
}

static inline int tri_test(const ray_t* r, uint32_t orig, float* t_out)
{
    const float* v = tri[orig];
    const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
    const float p[3] = {r->d[1] * e2[2] - r->d[2] * e2[1], r->d[2] * e2[0] - r->d[0] * e2[2], r->d[0] * e2[1] - r->d[1] * e2[0]};
    const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det < 1e-8f && det > -1e-8f) return 0;
    const float inv = 1.0f / det;
    const float tv[3] = {r->o[0] - v[0], r->o[1] - v[1], r->o[2] - v[2]};
    const float u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv;
    if (u < 0.0f || u > 1.0f) return 0;
    const float q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float w = (r->d[0] * q[0] + r->d[1] * q[1] + r->d[2] * q[2]) * inv;
    if (w < 0.0f || u + w > 1.0f) return 0;
    *t_out = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    return 1;
}

// walk_packet_lean's control flow for one 8x8 tile; returns the steps, fills best_t / best_leaf of the 64 lanes
static int g_tw = 8, g_th = 8;        // tile shape (g_tw x g_th = 64 lanes)
static int g_near_rule = 0;          // 0: majority vote (the product); 1: the child whose minimum entry distance is smaller
static int g_pop_nearest = 0;        // 1: a pop takes the waiting entry with the smallest entry distance, not the newest
static int g_steal = 0;
static int g_predict = 0;
static __thread uint32_t tl_step_cap = 0;       // if set: walk_tile stops after this many steps
static __thread uint32_t* tl_lane_work = NULL;   // if set: [64] steps in which the lane wanted a child or tested a leaf
static int g_cull = 0;               // 1: a popped node whose wave-minimum entry distance exceeds every active lane's best t is dropped unfetched
static uint64_t g_culled = 0;
static uint32_t* g_cnt;              // leaves under each node (evaluate fills it)
static uint64_t g_hist[40][2];       // packet steps by log2(leaves under the node): [..][0] all, [..][1] steps that entered nothing and tested no leaf
static uint32_t walk_tile(const tree_t* t, const ray_t* rays, const int* act, float* best_t, uint32_t* best_leaf, uint32_t* leaf_tests)
{
    uint32_t stack[256], sp = 0, steps = 0, node = 0;
    float stack_t[256];
    for (int l = 0; l < 64; l++) { best_t[l] = act[l] ? MAXF : -INFINITY; best_leaf[l] = 0xFFFFFFFFu; }
    for (;;) {
        const node_t* nd = &t->nd[node];
        if (tl_step_cap && steps >= tl_step_cap) break;
        steps++;
        const int lg = 31 - __builtin_clz(g_cnt[node]);
        int useful = 0;
        float tl[64], tr[64];
        uint64_t ml = 0, mr = 0;
        for (int l = 0; l < 64; l++) {
            const int bl = slab(&nd->lb, &rays[l], &tl[l]), br = slab(&nd->rb, &rays[l], &tr[l]);
            if (bl && !(tl[l] > best_t[l])) ml |= 1ull << l;
            if (br && !(tr[l] > best_t[l])) mr |= 1ull << l;
        }
        if (nd->l & LEAF) {
            if (ml) {
                (*leaf_tests)++; useful = 1;
                if (tl_lane_work) for (int l = 0; l < 64; l++) if ((ml >> l) & 1) tl_lane_work[l]++;
                const uint32_t pos = nd->l & ~LEAF;
                for (int l = 0; l < 64; l++) {
                    float tt;
                    if (((ml >> l) & 1) && tri_test(&rays[l], sidx[pos], &tt) &&
                        (tt < best_t[l] || (tt == best_t[l] && sidx[pos] < sidx[best_leaf[l] == 0xFFFFFFFFu ? pos : best_leaf[l]]))) {
                        best_t[l] = tt; best_leaf[l] = pos;
                    }
                    if (tr[l] > best_t[l]) mr &= ~(1ull << l);
                }
            }
            ml = 0;
        }
        if (nd->r & LEAF) {
            if (mr) {
                (*leaf_tests)++; useful = 1;
                if (tl_lane_work) for (int l = 0; l < 64; l++) if ((mr >> l) & 1) tl_lane_work[l]++;
                const uint32_t pos = nd->r & ~LEAF;
                for (int l = 0; l < 64; l++) {
                    float tt;
                    if (((mr >> l) & 1) && tri_test(&rays[l], sidx[pos], &tt) &&
                        (tt < best_t[l] || (tt == best_t[l] && sidx[pos] < sidx[best_leaf[l] == 0xFFFFFFFFu ? pos : best_leaf[l]]))) {
                        best_t[l] = tt; best_leaf[l] = pos;
                    }
                    if (tl[l] > best_t[l]) ml &= ~(1ull << l);
                }
            }
            mr = 0;
        }
        if (ml || mr) useful = 1;
        if (tl_lane_work) for (int l = 0; l < 64; l++) if (((ml | mr) >> l) & 1) tl_lane_work[l]++;
#pragma omp atomic
        g_hist[lg][0]++;
        if (!useful) {
#pragma omp atomic
            g_hist[lg][1]++;
        }
        if (ml && mr) {
            const uint64_t both = ml & mr;
            uint64_t le = 0;
            for (int l = 0; l < 64; l++) if (tl[l] <= tr[l]) le |= 1ull << l;
            const int by_votes = __builtin_popcountll(both & le) - __builtin_popcountll(both & ~le);
            const int by_lanes = __builtin_popcountll(ml) - __builtin_popcountll(mr);
            int l_near = (by_votes != 0 ? by_votes : by_lanes) >= 0;
            if (g_near_rule == 1) {
                float ml_min = INFINITY, mr_min = INFINITY;
                for (int l = 0; l < 64; l++) { if ((ml >> l) & 1) ml_min = fminf(ml_min, tl[l]); if ((mr >> l) & 1) mr_min = fminf(mr_min, tr[l]); }
                l_near = ml_min <= mr_min;
            }
            {   // entry distance of the far child: the minimum over the lanes that want it
                const uint64_t mf = l_near ? mr : ml;
                float m = INFINITY;
                for (int l = 0; l < 64; l++) if ((mf >> l) & 1) m = fminf(m, l_near ? tr[l] : tl[l]);
                stack_t[sp] = m;
            }
            stack[sp++] = l_near ? nd->r : nd->l;
            node = l_near ? nd->l : nd->r;
        } else if (ml) node = nd->l;
        else if (mr) node = nd->r;
        else {
            int done = 0;
            for (;;) {
                if (!sp) { done = 1; break; }
                if (g_pop_nearest) {                   // bring the nearest waiting entry to the top
                    uint32_t at = sp - 1;
                    for (uint32_t k = 0; k + 1 < sp; k++) if (stack_t[k] < stack_t[at]) at = k;
                    const uint32_t tn = stack[at]; const float tt = stack_t[at];
                    stack[at] = stack[sp - 1]; stack_t[at] = stack_t[sp - 1];
                    stack[sp - 1] = tn; stack_t[sp - 1] = tt;
                }
                --sp;
                if (g_cull) {
                    float worst = -INFINITY;            // the largest best t of the active lanes
                    for (int l = 0; l < 64; l++) if (act[l]) worst = fmaxf(worst, best_t[l]);
                    if (stack_t[sp] > worst) {
#pragma omp atomic
                        g_culled++;
                        continue;
                    }
                }
                node = stack[sp];
                break;
            }
            if (done) break;
        }
    }
    return steps;
}

// a single ray, near child first, t pruning, hits accepted with t > t_min: node visits
static uint32_t walk_ray(const tree_t* t, const ray_t* r, float t_min, float* best_out, uint32_t* leaf_out)
{
    uint32_t stack[256], sp = 0, visits = 0, node = 0;
    float best = MAXF; uint32_t best_leaf = 0xFFFFFFFFu;
    for (;;) {
        const node_t* nd = &t->nd[node];
        visits++;
        float tl, tr;
        int hl = slab(&nd->lb, r, &tl) && !(tl > best), hr = slab(&nd->rb, r, &tr) && !(tr > best);
        if ((nd->l & LEAF) && hl) {
            float tt;
            if (tri_test(r, sidx[nd->l & ~LEAF], &tt) && tt > t_min && tt < best) { best = tt; best_leaf = nd->l & ~LEAF; }
            hr = hr && !(tr > best);
        }
        if ((nd->r & LEAF) && hr) {
            float tt;
            if (tri_test(r, sidx[nd->r & ~LEAF], &tt) && tt > t_min && tt < best) { best = tt; best_leaf = nd->r & ~LEAF; }
            hl = hl && !(tl > best);
        }
        if (nd->l & LEAF) hl = 0;
        if (nd->r & LEAF) hr = 0;
        if (hl && hr) {
            const int l_near = tl <= tr;
            stack[sp++] = l_near ? nd->r : nd->l;
            node = l_near ? nd->l : nd->r;
        } else if (hl) node = nd->l;
        else if (hr) node = nd->r;
        else {
            if (!sp) break;
            node = stack[--sp];
        }
    }
    *best_out = best; *leaf_out = best_leaf;
    return visits;
}


// ---- wave-level work stealing for the per-ray walk (cfg5's later bounces; DESIGN 14.6, profiles/r5/e_work_stealing_per_ray_walk.txt) -----------------------------------------------
// 64 rays, one per lane, stepped in lockstep (a step = one node visit per active lane, the latency of one dependent fetch).
// Without stealing the wave takes as many steps as its longest ray.  With it, a lane whose ray is finished takes the OLDEST waiting
// entry (the bottom of the stack: the farthest subtree) of the lane with the most waiting entries and walks it for that ray,
// starting from the owner's best t at that moment; results are merged per ray (min t; ties to the lower triangle).
typedef struct { int active; uint32_t ray, node, sb, sp; float best; uint32_t best_leaf; uint32_t stack[128]; } lane_t;

static inline void lane_step(const tree_t* t, const ray_t* r, float t_min, lane_t* L)
{
    const node_t* nd = &t->nd[L->node];
    float tl, tr;
    int hl = slab(&nd->lb, r, &tl) && !(tl > L->best), hr = slab(&nd->rb, r, &tr) && !(tr > L->best);
    if ((nd->l & LEAF) && hl) {
        float tt;
        const uint32_t pos = nd->l & ~LEAF;
        if (tri_test(r, sidx[pos], &tt) && tt > t_min && (tt < L->best || (tt == L->best && sidx[pos] < sidx[L->best_leaf]))) { L->best = tt; L->best_leaf = pos; }
        hr = hr && !(tr > L->best);
    }
    if ((nd->r & LEAF) && hr) {
        float tt;
        const uint32_t pos = nd->r & ~LEAF;
        if (tri_test(r, sidx[pos], &tt) && tt > t_min && (tt < L->best || (tt == L->best && sidx[pos] < sidx[L->best_leaf]))) { L->best = tt; L->best_leaf = pos; }
        hl = hl && !(tl > L->best);
    }
    if (nd->l & LEAF) hl = 0;
    if (nd->r & LEAF) hr = 0;
    if (hl && hr) {
        const int l_near = tl <= tr;
        L->stack[L->sp++] = l_near ? nd->r : nd->l;
        L->node = l_near ? nd->l : nd->r;
    } else if (hl) L->node = nd->l;
    else if (hr) L->node = nd->r;
    else if (L->sp > L->sb) L->node = L->stack[--L->sp];
    else L->active = 0;
}

// returns the wave's steps; `steals_per_step` = 0: no stealing.  best[] / leaf[]: the rays' merged results
static uint32_t wave_walk(const tree_t* t, const ray_t* rays, int n_rays, float t_min, int steals_per_step, float* best, uint32_t* leaf, uint64_t* visits)
{
    lane_t L[64];
    for (int l = 0; l < 64; l++) {
        L[l].active = l < n_rays; L[l].ray = (uint32_t)l; L[l].node = 0; L[l].sb = L[l].sp = 0; L[l].best = MAXF; L[l].best_leaf = 0;
        if (l < n_rays) { best[l] = MAXF; leaf[l] = 0xFFFFFFFFu; }
    }
    uint32_t steps = 0;
    for (;;) {
        int any = 0;
        for (int l = 0; l < 64; l++) any |= L[l].active;
        if (!any) break;
        steps++;
        for (int l = 0; l < 64; l++) {
            if (!L[l].active) continue;
            (*visits)++;
            lane_step(t, &rays[L[l].ray], t_min, &L[l]);
            if (!L[l].active) {          // merge into the ray's result
                const uint32_t r = L[l].ray;
                if (L[l].best < best[r] || (L[l].best == best[r] && L[l].best < MAXF && sidx[L[l].best_leaf] < sidx[leaf[r]])) { best[r] = L[l].best; leaf[r] = L[l].best_leaf; }
            }
        }
        for (int s = 0; s < steals_per_step; s++) {
            int thief = -1, donor = -1; uint32_t depth = 0;
            for (int l = 0; l < 64; l++) {
                if (!L[l].active) { if (thief < 0) thief = l; }
                else if (L[l].sp - L[l].sb > depth) { depth = L[l].sp - L[l].sb; donor = l; }
            }
            if (thief < 0 || donor < 0 || depth < 1) break;
            L[thief].active = 1; L[thief].ray = L[donor].ray; L[thief].node = L[donor].stack[L[donor].sb++];
            L[thief].sb = L[thief].sp = 0; L[thief].best = L[donor].best; L[thief].best_leaf = L[donor].best_leaf;
        }
    }
    return steps;
}

static int cmp_u32(const void* a, const void* b) { return *(const uint32_t*)a < *(const uint32_t*)b ? -1 : *(const uint32_t*)a > *(const uint32_t*)b; }

static uint32_t pcg(uint32_t* s) { *s = *s * 747796405u + 2891336453u; uint32_t w = ((*s >> ((*s >> 28) + 4)) ^ *s) * 277803737u; return (w >> 22) ^ w; }

static uint32_t count_leaves(const tree_t* t, uint32_t node)
{
    const uint32_t c = ((t->nd[node].l & LEAF) ? 1u : count_leaves(t, t->nd[node].l)) + ((t->nd[node].r & LEAF) ? 1u : count_leaves(t, t->nd[node].r));
    g_cnt[node] = c;
    return c;
}

static int g_show_hist = 0;
static void evaluate(const char* name, const tree_t* t, double build_s)
{
    uint32_t depth;
    g_cnt = malloc((size_t)n * 4);
    count_leaves(t, 0);
    const double sah = tree_sah(t, &depth);
    printf("%-44s sah %8.2f depth %3u build %6.2fs |", name, sah, depth, build_s);
    const float cams[2] = {250.0f, 160.0f};
    for (int ci = 0; ci < 2; ci++) {
        camera_t c = {1920, 1080, tanf(30.0f * (float)M_PI / 180.0f), 0.3f, {0.0f, 0.0f, cams[ci]}};
        const uint32_t tx = (c.w + g_tw - 1) / g_tw, ty = (c.h + g_th - 1) / g_th, tiles = tx * ty;
        uint32_t* steps = malloc(tiles * 4);
        uint64_t total = 0, leaf_tests = 0, hits = 0; double tsum = 0.0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : total, leaf_tests, hits, tsum)
        for (uint32_t tile = 0; tile < tiles; tile++) {
            ray_t rays[64]; int act[64]; float bt[64]; uint32_t bl[64];
            for (int l = 0; l < 64; l++) {
                const uint32_t px = (tile % tx) * g_tw + (l % g_tw), py = (tile / tx) * g_th + (l / g_tw);
                act[l] = px < (uint32_t)c.w && py < (uint32_t)c.h;
                rays[l] = camera_ray(&c, px < (uint32_t)c.w ? px : c.w - 1, py < (uint32_t)c.h ? py : c.h - 1);
            }
            uint32_t lt = 0;
            steps[tile] = walk_tile(t, rays, act, bt, bl, &lt);
            total += steps[tile]; leaf_tests += lt;
            for (int l = 0; l < 64; l++) if (act[l] && bt[l] < MAXF) { hits++; tsum += bt[l]; }
        }
        if (g_cull) { printf(" [culled at pop: %llu]", (unsigned long long)g_culled); g_culled = 0; }
        if (g_predict) {
            // how well do a few single rays per tile predict which tiles are heavy?  predictor = node visits of the per-ray walk
            // for the tile's centre ray / the largest over a 2x2 / 4x4 grid of its rays; recall of the tiles of >= 256 packet steps
            // among the 506 (1/64 of the frame) the predictor ranks highest, and among twice / four times as many
            const int grids[3] = {1, 2, 4};
            for (int gi = 0; gi < 3; gi++) {
                const int gdim = grids[gi];
                uint32_t* pred = malloc(tiles * 4);
#pragma omp parallel for schedule(dynamic, 16)
                for (uint32_t tile = 0; tile < tiles; tile++) {
                    uint32_t m = 0;
                    for (int a = 0; a < gdim; a++)
                        for (int b = 0; b < gdim; b++) {
                            uint32_t px = (tile % tx) * g_tw + (uint32_t)((2 * a + 1) * g_tw / (2 * gdim)), py = (tile / tx) * g_th + (uint32_t)((2 * b + 1) * g_th / (2 * gdim));
                            if (px >= (uint32_t)c.w) px = c.w - 1;
                            if (py >= (uint32_t)c.h) py = c.h - 1;
                            ray_t r = camera_ray(&c, px, py);
                            float bt; uint32_t bl;
                            const uint32_t v = walk_ray(t, &r, 0.0f, &bt, &bl);
                            if (v > m) m = v;
                        }
                    pred[tile] = m;
                }
                uint32_t* sorted = malloc(tiles * 4);
                memcpy(sorted, pred, tiles * 4);
                qsort(sorted, tiles, 4, cmp_u32);
                uint32_t heavy = 0;
                for (uint32_t i = 0; i < tiles; i++) heavy += steps[i] >= 256;
                printf("\n   predictor %dx%d rays per tile: heavy tiles %u;", gdim, gdim, heavy);
                for (int mult = 1; mult <= 4; mult *= 2) {
                    const uint32_t K = (tiles / 64) * mult, thr = sorted[tiles - K];
                    uint32_t found = 0, picked = 0;
                    for (uint32_t i = 0; i < tiles; i++) if (pred[i] >= thr) { picked++; found += steps[i] >= 256; }
                    printf(" top %u (picked %u): recall %.2f;", K, picked, (double)found / heavy);
                }
                free(pred); free(sorted);
            }
            printf("\n");
            // the same from COARSE PACKETS: 64 rays = 4 x 4 tiles x (2 x 2 rays per tile), walked as one packet (the product's walk);
            // a tile's predictor = the largest per-lane work count of its 4 rays.  What the pre-pass costs: its steps and its longest packet
            if (g_tw == 8 && g_th == 8)
            for (uint32_t cap = 0; cap <= 192; cap = cap ? cap * 2 : 48) {
                const uint32_t cx = (tx + 3) / 4, cy = (ty + 3) / 4, coarse = cx * cy;
                uint32_t* pred = calloc(tiles, 4);
                uint64_t csteps = 0; uint32_t cmax = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : csteps) reduction(max : cmax)
                for (uint32_t cp = 0; cp < coarse; cp++) {
                    ray_t rays[64]; int act[64]; float bt[64]; uint32_t bl[64]; uint32_t work[64] = {0};
                    uint32_t tile_of[64];
                    for (int l = 0; l < 64; l++) {
                        // lane l: tile (l & 3, (l >> 2) & 3) of the block, ray (l >> 4) & 1, (l >> 5) & 1 of its 2 x 2
                        const uint32_t bx = (cp % cx) * 4 + (l & 3), by = (cp / cx) * 4 + ((l >> 2) & 3);
                        const uint32_t px = bx * 8 + 2 + 4 * ((l >> 4) & 1), py = by * 8 + 2 + 4 * ((l >> 5) & 1);
                        act[l] = bx < tx && by < ty && px < (uint32_t)c.w && py < (uint32_t)c.h;
                        tile_of[l] = act[l] ? by * tx + bx : 0xFFFFFFFFu;
                        rays[l] = camera_ray(&c, px < (uint32_t)c.w ? px : c.w - 1, py < (uint32_t)c.h ? py : c.h - 1);
                    }
                    uint32_t lt = 0;
                    tl_lane_work = work;
                    tl_step_cap = cap;
                    const uint32_t st = walk_tile(t, rays, act, bt, bl, &lt);
                    tl_lane_work = NULL;
                    tl_step_cap = 0;
                    csteps += st; if (st > cmax) cmax = st;
                    for (int l = 0; l < 64; l++) if (tile_of[l] != 0xFFFFFFFFu) {
#pragma omp critical
                        if (work[l] > pred[tile_of[l]]) pred[tile_of[l]] = work[l];
                    }
                }
                uint32_t* sorted = malloc(tiles * 4);
                memcpy(sorted, pred, tiles * 4);
                qsort(sorted, tiles, 4, cmp_u32);
                uint32_t heavy = 0;
                for (uint32_t i = 0; i < tiles; i++) heavy += steps[i] >= 256;
                printf("   coarse pre-pass (step cap %u): %u packets, %llu steps (%.1f per packet, longest %u);", cap, coarse, (unsigned long long)csteps, (double)csteps / coarse, cmax);
                for (int mult = 1; mult <= 4; mult *= 2) {
                    const uint32_t K = (tiles / 64) * mult, thr = sorted[tiles - K];
                    uint32_t found = 0, picked = 0;
                    for (uint32_t i = 0; i < tiles; i++) if (pred[i] >= thr) { picked++; found += steps[i] >= 256; }
                    printf(" top %u (picked %u): recall %.2f;", K, picked, (double)found / heavy);
                }
                printf("\n");
                free(pred); free(sorted);
            }
        }
        qsort(steps, tiles, 4, cmp_u32);
        uint32_t heavy = 0;
        for (uint32_t i = 0; i < tiles; i++) heavy += steps[i] >= 256;
        printf(" cam%d: steps %8llu (%.1f/tile, p50 %u p99 %u max %u, >=256: %u) leaf %llu hits %llu tsum %.6g |", ci,
               (unsigned long long)total, (double)total / tiles, steps[tiles / 2], steps[tiles * 99 / 100], steps[tiles - 1], heavy,
               (unsigned long long)leaf_tests, (unsigned long long)hits, tsum);
        free(steps);
        if (g_show_hist) {
            printf("\n   steps by log2(leaves under the node) [all / useless]:");
            for (int k = 1; k < 21; k++) printf(" %d:%llu/%llu", k, (unsigned long long)g_hist[k][0], (unsigned long long)g_hist[k][1]);
            printf("\n");
        }
        memset(g_hist, 0, sizeof g_hist);
    }
    // diffuse bounce rays from the primary hits of every 4th pixel (camera 0)
    {
        camera_t c = {1920, 1080, tanf(30.0f * (float)M_PI / 180.0f), 0.3f, {0.0f, 0.0f, 250.0f}};
        uint64_t pv = 0, bv = 0, rays = 0, b2 = 0;
        uint32_t* vis = calloc((size_t)(c.w / 4) * (c.h / 4), 4);      // node visits of every bounce-1 ray (0: the path had ended)
        ray_t* brays = malloc((size_t)(c.w / 4) * (c.h / 4) * sizeof(ray_t));   // the bounce-1 rays themselves (for the wave simulation)
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : pv, bv, rays, b2)
        for (int py = 0; py < c.h; py += 4)
            for (int px = 0; px < c.w; px += 4) {
                ray_t r = camera_ray(&c, px, py);
                float bt; uint32_t bl;
                pv += walk_ray(t, &r, 0.0f, &bt, &bl);
                uint32_t seed = (uint32_t)(py * c.w + px) * 9781u + 1u;
                for (int bounce = 0; bounce < 2 && bl != 0xFFFFFFFFu; bounce++) {
                    const float* v = tri[sidx[bl]];
                    const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
                    float nrm[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                    float len = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
                    if (!(len > 0.0f)) break;
                    float side = nrm[0] * r.d[0] + nrm[1] * r.d[1] + nrm[2] * r.d[2];
                    for (int k = 0; k < 3; k++) nrm[k] = (side > 0.0f ? -nrm[k] : nrm[k]) / len;
                    // uniform point on the sphere
                    const float z = 1.0f - 2.0f * ((float)(pcg(&seed) >> 8) / 16777216.0f), ph = 6.2831853f * ((float)(pcg(&seed) >> 8) / 16777216.0f);
                    const float s = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                    float d[3] = {nrm[0] + s * cosf(ph), nrm[1] + s * sinf(ph), nrm[2] + z};
                    len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    if (!(len > 1e-6f)) break;
                    ray_t q;
                    for (int k = 0; k < 3; k++) { q.o[k] = r.o[k] + r.d[k] * bt; q.d[k] = d[k] / len; q.inv[k] = 1.0f / q.d[k]; }
                    const uint32_t visits_ = walk_ray(t, &q, 1e-3f, &bt, &bl);
                    if (bounce == 0) { bv += visits_; rays++; vis[(size_t)(py / 4) * (c.w / 4) + px / 4] = visits_; brays[(size_t)(py / 4) * (c.w / 4) + px / 4] = q; } else b2 += visits_;
                    r = q;
                }
            }
        {
            const size_t nv = (size_t)(c.w / 4) * (c.h / 4);
            // a wave of 64 consecutive live rays takes as long as its longest ray: mean of the per-wave maxima against the mean ray
            uint64_t wave_max_sum = 0, waves = 0; uint32_t in_wave = 0, wmax = 0, gmax = 0;
            for (size_t i = 0; i < nv; i++) {
                if (!vis[i]) continue;
                if (vis[i] > wmax) wmax = vis[i];
                if (vis[i] > gmax) gmax = vis[i];
                if (++in_wave == 64) { wave_max_sum += wmax; waves++; in_wave = 0; wmax = 0; }
            }
            if (g_steal) {
                // the live bounce-1 rays in pixel order, 64 (or 32: half-filled waves, as the later bounces' launches) per wave
                ray_t* live_rays = malloc(nv * sizeof(ray_t));
                size_t nl = 0;
                for (size_t i = 0; i < nv; i++) if (vis[i]) live_rays[nl++] = brays[i];
                for (int per_wave = 64; per_wave >= 32; per_wave /= 2) {
                    for (int sps = 0; sps <= 4; sps = sps ? sps * 2 : 1) {
                        uint64_t steps_sum = 0, visits = 0, worst = 0, mismatch = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : steps_sum, visits, mismatch) reduction(max : worst)
                        for (size_t w0 = 0; w0 < nl; w0 += (size_t)per_wave) {
                            const int cnt = (int)(nl - w0 < (size_t)per_wave ? nl - w0 : (size_t)per_wave);
                            float best[64]; uint32_t leaf[64]; uint64_t v = 0;
                            const uint32_t st = wave_walk(t, live_rays + w0, cnt, 1e-3f, sps, best, leaf, &v);
                            steps_sum += st; visits += v; if (st > worst) worst = st;
                            for (int l = 0; l < cnt; l++) {
                                float bt; uint32_t bl;
                                walk_ray(t, &live_rays[w0 + l], 1e-3f, &bt, &bl);
                                if (bt != best[l]) mismatch++;
                            }
                        }
                        printf("\n   waves of %d rays, %d steal(s) per step: longest wave %llu steps, mean %.1f, node visits %llu, results differing from the lone walk: %llu",
                               per_wave, sps, (unsigned long long)worst, (double)steps_sum / (double)((nl + per_wave - 1) / per_wave), (unsigned long long)visits,
                               (unsigned long long)mismatch);
                    }
                }
                printf("\n");
                free(live_rays);
            }
            free(brays);
            qsort(vis, nv, 4, cmp_u32);
            size_t first = 0; while (first < nv && !vis[first]) first++;
            const size_t live = nv - first;
            printf(" [bounce-1 visits: p50 %u p90 %u p99 %u max %u; mean of per-wave (64 rays) maxima %.1f]", vis[first + live / 2], vis[first + live * 9 / 10],
                   vis[first + live * 99 / 100], gmax, waves ? (double)wave_max_sum / waves : 0.0);
            free(vis);
        }
        printf(" per-ray: primary %.1f bounce1 %.1f (x%llu) bounce2 sum %llu\n", (double)pv / (c.w / 4 * (c.h / 4)), (double)bv / (double)rays,
               (unsigned long long)rays, (unsigned long long)b2);
    }
    fflush(stdout);
    free(g_cnt);
}

// ---- k-wide collapses of a binary tree for the per-ray walk (round 6, VERDICT r5 item 7: decide cfg5's 8-wide node HERE) -------------------
// A wide node = a binary node with its children opened by surface — the internal child with the largest box area is replaced by its
// two children — until it has k children or only leaves are left.  k = 4 is the product's four-wide form (lbvh_path.hip: up to four
// child boxes per 128-byte line).  k = 8 fits one 128-byte line only with boxes on a node-local 8-bit grid (origin + extent of the
// node: 24 B, 8 x 6 B of quantised corners, 8 x 4 B of references = 104 B): `quant` rounds every child box OUTWARD onto that grid,
// as such a node would have to.  Counted per diffuse bounce ray: node lines fetched (one per wide node visited) and triangle lines.
typedef struct { uint32_t n; uint32_t child[8]; box_t box[8]; } wnode_t;
typedef struct { wnode_t* nd; uint32_t count; } wtree_t;

static box_t quantise_outward(box_t b, box_t frame)
{
    box_t q;
    for (int k = 0; k < 3; k++) {
        const float ext = frame.mx[k] - frame.mn[k], step = ext > 0.0f ? ext / 255.0f : 1.0f;
        float lo = floorf((b.mn[k] - frame.mn[k]) / step), hi = ceilf((b.mx[k] - frame.mn[k]) / step);
        lo = fminf(fmaxf(lo, 0.0f), 255.0f); hi = fminf(fmaxf(hi, 0.0f), 255.0f);
        q.mn[k] = frame.mn[k] + lo * step; q.mx[k] = frame.mn[k] + hi * step;
        if (q.mn[k] > b.mn[k]) q.mn[k] = b.mn[k];          // (rounding of the products: never inward)
        if (q.mx[k] < b.mx[k]) q.mx[k] = b.mx[k];
    }
    return q;
}

static uint32_t collapse_node(const tree_t* t, wtree_t* w, uint32_t bin, uint32_t k, int quant)
{
    const uint32_t me = w->count++;
    uint32_t ch[8]; box_t bx[8]; uint32_t cn = 2;
    ch[0] = t->nd[bin].l; bx[0] = t->nd[bin].lb; ch[1] = t->nd[bin].r; bx[1] = t->nd[bin].rb;
    while (cn < k) {
        int pick = -1; float area = -1.0f;
        for (uint32_t i = 0; i < cn; i++) if (!(ch[i] & LEAF) && box_area(bx[i]) > area) { area = box_area(bx[i]); pick = (int)i; }
        if (pick < 0) break;
        const node_t* c = &t->nd[ch[pick]];
        ch[pick] = c->l; bx[pick] = c->lb;
        ch[cn] = c->r; bx[cn] = c->rb; cn++;
    }
    box_t frame = box_empty();
    for (uint32_t i = 0; i < cn; i++) frame = box_union(frame, bx[i]);
    wnode_t nd; nd.n = cn;
    for (uint32_t i = 0; i < cn; i++) {
        nd.box[i] = quant ? quantise_outward(bx[i], frame) : bx[i];
        nd.child[i] = (ch[i] & LEAF) ? ch[i] : collapse_node(t, w, ch[i], k, quant);
    }
    w->nd[me] = nd;
    return me;
}

// near-first per-ray walk over wide nodes: node lines fetched; *tris = triangle lines fetched (own-box test passed)
static uint32_t walk_wide(const wtree_t* w, const ray_t* r, float t_min, float* best_out, uint32_t* leaf_out, uint32_t* tris)
{
    struct { uint32_t node; float t; } stack[512];
    uint32_t sp = 0, visits = 0, node = 0;
    float best = MAXF; uint32_t best_leaf = 0xFFFFFFFFu;
    for (;;) {
        const wnode_t* nd = &w->nd[node];
        visits++;
        uint32_t hit_i[8]; float hit_t[8]; uint32_t hn = 0;
        for (uint32_t i = 0; i < nd->n; i++) {
            float te;
            if (!slab(&nd->box[i], r, &te) || te > best) continue;
            if (nd->child[i] & LEAF) {
                float tt;
                (*tris)++;
                if (tri_test(r, sidx[nd->child[i] & ~LEAF], &tt) && tt > t_min && tt < best) { best = tt; best_leaf = nd->child[i] & ~LEAF; }
            } else { hit_i[hn] = nd->child[i]; hit_t[hn] = te; hn++; }
        }
        // far to near onto the stack, the nearest is walked next (entries beyond the best hit by now are dropped when popped)
        for (uint32_t a = 1; a < hn; a++)
            for (uint32_t b = a; b > 0 && hit_t[b] > hit_t[b - 1]; b--) {
                const float tf = hit_t[b]; hit_t[b] = hit_t[b - 1]; hit_t[b - 1] = tf;
                const uint32_t ti = hit_i[b]; hit_i[b] = hit_i[b - 1]; hit_i[b - 1] = ti;
            }
        for (uint32_t a = 0; a < hn; a++) if (!(hit_t[a] > best)) { stack[sp].node = hit_i[a]; stack[sp].t = hit_t[a]; sp++; }
        for (;;) {
            if (!sp) { *best_out = best; *leaf_out = best_leaf; return visits; }
            sp--;
            if (!(stack[sp].t > best)) { node = stack[sp].node; break; }
        }
    }
}

static int g_wide = 0;
// diffuse bounce rays (the primary hits of every 4th pixel of cfg2's camera, two bounces) over the binary tree and its collapses
static void evaluate_wide(const tree_t* t)
{
    const struct { uint32_t k; int quant; const char* name; } forms[] = {
        {2, 0, "binary (2 child boxes per 64-byte line)"}, {4, 0, "4-wide, fp32 boxes (the product's walkers: one 128-byte line)"},
        {8, 0, "8-wide, fp32 boxes (would be two lines: 224 B)"}, {8, 1, "8-wide, node-local 8-bit grid (one 128-byte line)"},
        {6, 0, "6-wide, fp32 boxes (would be 168 B)"}, {6, 1, "6-wide, node-local 8-bit grid"}, {4, 1, "4-wide, node-local 8-bit grid (64-byte line)"}};
    camera_t c = {1920, 1080, tanf(30.0f * (float)M_PI / 180.0f), 0.3f, {0.0f, 0.0f, 250.0f}};
    printf("   per-ray walks over collapses of this tree: rays = primary hits of every 4th pixel + two diffuse bounces (as evaluate())\n");
    for (size_t f = 0; f < sizeof forms / sizeof forms[0]; f++) {
        wtree_t w; w.nd = malloc((size_t)(n - 1) * sizeof(wnode_t)); w.count = 0;
        collapse_node(t, &w, 0, forms[f].k, forms[f].quant);
        uint64_t pv = 0, pt = 0, bv[2] = {0, 0}, bt_[2] = {0, 0}, rays[2] = {0, 0}, mism = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : pv, pt, bv[:2], bt_[:2], rays[:2], mism)
        for (int py = 0; py < c.h; py += 4)
            for (int px = 0; px < c.w; px += 4) {
                ray_t r = camera_ray(&c, px, py);
                float bt; uint32_t bl, tl = 0;
                pv += walk_wide(&w, &r, 0.0f, &bt, &bl, &tl); pt += tl;
                { float b2; uint32_t l2; walk_ray(t, &r, 0.0f, &b2, &l2); if (b2 != bt) mism++; }
                uint32_t seed = (uint32_t)(py * c.w + px) * 9781u + 1u;
                for (int bounce = 0; bounce < 2 && bl != 0xFFFFFFFFu; bounce++) {
                    const float* v = tri[sidx[bl]];
                    const float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
                    float nrm[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                    float len = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
                    if (!(len > 0.0f)) break;
                    float side = nrm[0] * r.d[0] + nrm[1] * r.d[1] + nrm[2] * r.d[2];
                    for (int k = 0; k < 3; k++) nrm[k] = (side > 0.0f ? -nrm[k] : nrm[k]) / len;
                    const float z = 1.0f - 2.0f * ((float)(pcg(&seed) >> 8) / 16777216.0f), ph = 6.2831853f * ((float)(pcg(&seed) >> 8) / 16777216.0f);
                    const float s_ = sqrtf(fmaxf(0.0f, 1.0f - z * z));
                    float d[3] = {nrm[0] + s_ * cosf(ph), nrm[1] + s_ * sinf(ph), nrm[2] + z};
                    len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    if (!(len > 1e-6f)) break;
                    ray_t q;
                    for (int k = 0; k < 3; k++) { q.o[k] = r.o[k] + r.d[k] * bt; q.d[k] = d[k] / len; q.inv[k] = 1.0f / q.d[k]; }
                    tl = 0;
                    bv[bounce] += walk_wide(&w, &q, 1e-3f, &bt, &bl, &tl); bt_[bounce] += tl; rays[bounce]++;
                    { float b2; uint32_t l2; walk_ray(t, &q, 1e-3f, &b2, &l2); if (b2 != bt) mism++; }
                    r = q;
                }
            }
        const double all_rays = (double)(c.w / 4) * (c.h / 4) + (double)rays[0] + (double)rays[1];
        printf("   %-66s nodes %8u | node lines per ray: primary %5.1f  bounce 1 %5.1f  bounce 2 %5.1f  all %5.1f | triangle lines per ray %4.2f | t differs from the binary walk: %llu\n",
               forms[f].name, w.count, (double)pv / ((double)(c.w / 4) * (c.h / 4)), (double)bv[0] / (double)rays[0], (double)bv[1] / (double)rays[1],
               (double)(pv + bv[0] + bv[1]) / all_rays, (double)(pt + bt_[0] + bt_[1]) / all_rays, (unsigned long long)mism);
        free(w.nd);
    }
    fflush(stdout);
}

static tree_t new_tree(void) { tree_t t; t.nd = malloc((size_t)(n - 1) * sizeof(node_t)); t.count = 0; return t; }

static void run_range(const char* name, const uint32_t* keys, int top, int bottom, uint32_t threshold)
{
    tree_t t = new_tree();
    g_keys = keys; g_top_mode = top; g_bottom_mode = bottom; g_threshold = threshold;
    const double t0 = omp_get_wtime();
    box_t root;
    build_range(&t, 0, n - 1, &root);
    if (g_wide) evaluate_wide(&t);
    else evaluate(name, &t, omp_get_wtime() - t0);
    free(t.nd);
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: treelab scene.tri [what ...]\n"); return 2; }
    load_scene(argv[1]);
    g_suffix = malloc((size_t)(n + 1) * 4);
    uint32_t distinct = 1;
    for (uint32_t i = 1; i < n; i++) distinct += mkeys[i] != mkeys[i - 1];
    printf("%u triangles, %u distinct Morton codes, %d threads\n", n, distinct, omp_get_max_threads());
    char name[128];
    for (int a = 2; a < argc; a++) {
        const char* w = argv[a];
        if (!strcmp(w, "hist")) g_show_hist = 1;
        else if (!strcmp(w, "cull")) g_cull = 1;
        else if (!strcmp(w, "predict")) g_predict = 1;
        else if (!strcmp(w, "steal")) g_steal = 1;
        else if (!strcmp(w, "wide")) g_wide = 1;          // the following trees: per-ray walks over their 2 / 4 / 6 / 8-wide collapses only
        else if (!strcmp(w, "popnearest")) g_pop_nearest = 1;
        else if (!strncmp(w, "tile", 4)) { sscanf(w + 4, "%dx%d", &g_tw, &g_th); printf("tile %dx%d\n", g_tw, g_th); }
        else if (!strncmp(w, "near", 4)) { g_near_rule = atoi(w + 4); printf("near rule %d\n", g_near_rule); }
        else if (!strcmp(w, "radix")) run_range("radix(aligned keys) [product]", akeys, SPLIT_RADIX, SPLIT_RADIX, 0);
        else if (!strcmp(w, "sweep")) run_range("sweep SAH in sorted order, all levels", akeys, SPLIT_SWEEP, SPLIT_SWEEP, 0);
        else if (!strcmp(w, "median")) run_range("median split in sorted order", akeys, SPLIT_MEDIAN, SPLIT_MEDIAN, 0);
        else if (!strncmp(w, "top", 3)) {          // topN: sweep SAH for ranges > N leaves, radix below
            const uint32_t th = (uint32_t)atoi(w + 3);
            snprintf(name, sizeof name, "sweep SAH above %u leaves, radix below", th);
            run_range(name, akeys, SPLIT_SWEEP, SPLIT_RADIX, th);
        } else if (!strncmp(w, "bot", 3)) {        // botN: radix for ranges > N leaves, sweep SAH below
            const uint32_t th = (uint32_t)atoi(w + 3);
            snprintf(name, sizeof name, "radix above %u leaves, sweep SAH below", th);
            run_range(name, akeys, SPLIT_RADIX, SPLIT_SWEEP, th);
        } else if (!strncmp(w, "rsah", 4)) {       // rsahL: best of the radix splits of L levels, everywhere
            g_cand_levels = (uint32_t)atoi(w + 4);
            snprintf(name, sizeof name, "best of %u levels of radix splits (SAH)", g_cand_levels);
            run_range(name, akeys, SPLIT_RADIX_SAH, SPLIT_RADIX_SAH, 0);
        } else if (!strcmp(w, "binned")) {
            tree_t t = new_tree();
            g_perm = malloc((size_t)n * 4);
            for (uint32_t i = 0; i < n; i++) g_perm[i] = i;
            const double t0 = omp_get_wtime();
            box_t root;
            build_binned(&t, 0, n - 1, &root);
            evaluate("binned SAH, free leaf order (ceiling)", &t, omp_get_wtime() - t0);
            free(t.nd); free(g_perm);
        } else if (!strncmp(w, "ploc", 4)) {
            tree_t t = new_tree();
            const uint32_t radius = (uint32_t)atoi(w + 4);
            const double t0 = omp_get_wtime();
            build_ploc(&t, radius ? radius : 16);
            snprintf(name, sizeof name, "PLOC over the Morton order, radius %u", radius ? radius : 16);
            evaluate(name, &t, omp_get_wtime() - t0);
            free(t.nd);
        } else fprintf(stderr, "unknown: %s\n", w);
    }
    return 0;
}
