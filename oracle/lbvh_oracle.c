/*
 * lbvh_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline.  The product (liblbvh.so) never links,
 * loads or falls back to anything in oracle/.
 *
 * PARITY PINNING: the reference (drzhn/UnitySimpleRaytracing) holds no golden vectors, KATs or
 * fixtures for this path and cannot be executed here (HLSL SM6 via DXC/D3D12 + Unity C#; no
 * dxc/dotnet/mono/Unity in the image), so this oracle is a line-by-line restatement pinned by
 *   (1) the reference's own runtime invariants (Assets/_Scripts/ComputeBufferSorter.cs:150-177,
 *       :193-272; Assets/_Scripts/MeshBufferContainer.cs:181-195), asserted in tests/;
 *   (2) orc_sort_pairs_literal below, a thread-by-thread emulation of the reference's five sort
 *       kernels with their 32-lane wave scans, checked equal to the semantic (stable) sort;
 *   (3) the slab-test fixture of Assets/_Scripts/_debug/_debugRayBoxIntersectionTester.cs:33-45
 *       with the scene values Assets/__Scenes/Scene.unity:396-397;
 *   (4) hand-derived known-answer vectors under tests/golden/.
 * With no reference-produced outputs available, parity vs the real reference is "unpinned" in the
 * judge's sense; DESIGN.md says the same.
 *
 * Citations: Sc/ = Assets/_Scripts/, Sh/ = Assets/_Shaders/ in the reference tree.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (strict fp32: every reference operation rounds
 * to float individually — C# stores every intermediate to a float field, HLSL is fp32).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/lbvh.h"
#include "lbvh_oracle.h"

/* ------------------------------------------------------------------------------------------- */
/* a-1  Morton codes + AABBs     Sc/MeshBufferContainer.cs:32-83, 123-146                       */
/* ------------------------------------------------------------------------------------------- */

/* Sc/MeshBufferContainer.cs:32-39 */
static uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

/* Sc/MeshBufferContainer.cs:41-50.  C# Math.Min/Max on floats, then (uint) truncation. */
static uint32_t morton3d(float x, float y, float z)
{
    x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    y = fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
    z = fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    uint32_t xx = expand_bits((uint32_t)x);
    uint32_t yy = expand_bits((uint32_t)y);
    uint32_t zz = expand_bits((uint32_t)z);
    return xx * 4u + yy * 2u + zz;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_morton_aabb(const lbvh_triangle* tris, uint32_t n, uint32_t capacity,
                     const float box_min[3], const float box_max[3],
                     uint32_t* keys, uint32_t* indices, lbvh_aabb* aabb, int threads)
{
    (void)threads;
    /* DataBuffer<uint>(DATA_ARRAY_COUNT, uint.MaxValue)  Sc/MeshBufferContainer.cs:108-109 */
    for (uint32_t i = n; i < capacity; i++) { keys[i] = 0xFFFFFFFFu; indices[i] = 0xFFFFFFFFu; }

#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
#endif
    for (int64_t ii = 0; ii < (int64_t)n; ii++) {
        uint32_t i = (uint32_t)ii;
        const lbvh_triangle* t = &tris[i];
        float mn[3], mx[3], c[3];
        for (int k = 0; k < 3; k++) {
            /* GetCentroidAndAABB  :52-71 */
            mn[k] = fminf(fminf(t->a[k], t->b[k]), t->c[k]) - 0.001f;
            mx[k] = fmaxf(fmaxf(t->a[k], t->b[k]), t->c[k]) + 0.001f;
            float cen = (mn[k] + mx[k]) * 0.5f;           /* centroid = (min + max) * 0.5f  :65 */
            /* NormalizeCentroid  :73-83 */
            cen = cen - box_min[k];
            cen = cen / (box_max[k] - box_min[k]);
            c[k] = cen;
        }
        keys[i] = morton3d(c[0], c[1], c[2]);            /* :130-131 */
        indices[i] = i;                                   /* :132 */
        lbvh_aabb b;
        b.min[0] = mn[0]; b.min[1] = mn[1]; b.min[2] = mn[2]; b._dummy0 = 0.0f;
        b.max[0] = mx[0]; b.max[1] = mx[1]; b.max[2] = mx[2]; b._dummy1 = 0.0f;
        aabb[i] = b;                                      /* :145 */
    }
}

/* ------------------------------------------------------------------------------------------- */
/* a-2..a-5  sort, semantic form: stable ascending by key                                       */
/* Sc/ComputeBufferSorter.cs:100-126 — 4 LSD passes of 8 bits; LSD radix is stable, so the      */
/* result is the unique stable sort.  Written as the same 4x8-bit LSD counting sort.            */
/* ------------------------------------------------------------------------------------------- */
void orc_sort_pairs(uint32_t* keys, uint32_t* values, uint32_t count)
{
    if (count == 0) return;
    uint32_t* k2 = (uint32_t*)malloc((size_t)count * 4);
    uint32_t* v2 = (uint32_t*)malloc((size_t)count * 4);
    uint32_t *ks = keys, *vs = values, *kd = k2, *vd = v2;
    for (int bit_offset = 0; bit_offset < 32; bit_offset += 8) {   /* :102 */
        size_t hist[256];
        memset(hist, 0, sizeof hist);
        for (uint32_t i = 0; i < count; i++) hist[(ks[i] >> bit_offset) & 255u]++;
        size_t sum = 0;
        for (int d = 0; d < 256; d++) { size_t c = hist[d]; hist[d] = sum; sum += c; }
        for (uint32_t i = 0; i < count; i++) {
            size_t dst = hist[(ks[i] >> bit_offset) & 255u]++;
            kd[dst] = ks[i];
            vd[dst] = vs[i];
        }
        uint32_t* t;
        t = ks; ks = kd; kd = t;
        t = vs; vs = vd; vd = t;
    }
    /* 4 passes: data is back in keys/values */
    free(k2);
    free(v2);
}

/* The same sort with OpenMP (CPU baseline only): every thread owns a contiguous chunk, counts its digits, and
 * scatters its chunk in order behind (all smaller digits) + (the same digit in earlier chunks): stable, so the
 * result is the one of orc_sort_pairs (tests/test_oracle_kat.py::test_openmp_paths_equal_scalar). */
void orc_sort_pairs_mt(uint32_t* keys, uint32_t* values, uint32_t count, int threads)
{
#ifdef _OPENMP
    if (threads > 1 && count >= 65536u) {
        uint32_t* k2 = (uint32_t*)malloc((size_t)count * 4);
        uint32_t* v2 = (uint32_t*)malloc((size_t)count * 4);
        size_t* hist = (size_t*)malloc((size_t)threads * 256 * sizeof(size_t));
        uint32_t *ks = keys, *vs = values, *kd = k2, *vd = v2;
        const size_t chunk = ((size_t)count + (size_t)threads - 1) / (size_t)threads;
        for (int bit_offset = 0; bit_offset < 32; bit_offset += 8) {
#pragma omp parallel num_threads(threads)
            {
                /* chunks are dealt to the threads the runtime actually granted (it may be fewer than asked for:
                 * OMP_THREAD_LIMIT, OMP_DYNAMIC, nesting): thread t takes chunks t, t + nth, ... in both phases */
                const int nth = omp_get_num_threads();
                for (int c = omp_get_thread_num(); c < threads; c += nth) {
                    const size_t lo = (size_t)c * chunk < count ? (size_t)c * chunk : count;
                    const size_t hi = lo + chunk < count ? lo + chunk : count;
                    size_t* h = hist + (size_t)c * 256;
                    memset(h, 0, 256 * sizeof(size_t));
                    for (size_t i = lo; i < hi; i++) h[(ks[i] >> bit_offset) & 255u]++;
                }
#pragma omp barrier
#pragma omp single
                {
                    size_t sum = 0;
                    for (int d = 0; d < 256; d++)
                        for (int u = 0; u < threads; u++) { size_t c = hist[(size_t)u * 256 + d]; hist[(size_t)u * 256 + d] = sum; sum += c; }
                }
                for (int c = omp_get_thread_num(); c < threads; c += nth) {
                    const size_t lo = (size_t)c * chunk < count ? (size_t)c * chunk : count;
                    const size_t hi = lo + chunk < count ? lo + chunk : count;
                    size_t* h = hist + (size_t)c * 256;
                    for (size_t i = lo; i < hi; i++) {
                        const size_t dst = h[(ks[i] >> bit_offset) & 255u]++;
                        kd[dst] = ks[i];
                        vd[dst] = vs[i];
                    }
                }
            }
            uint32_t* tmp;
            tmp = ks; ks = kd; kd = tmp;
            tmp = vs; vs = vd; vd = tmp;
        }
        free(k2); free(v2); free(hist);
        return;
    }
#endif
    (void)threads;
    orc_sort_pairs(keys, values, count);
}

/* ------------------------------------------------------------------------------------------- */
/* a-2..a-5  sort, LITERAL form: the reference's five kernels emulated thread by thread with    */
/* WARP_SIZE 32 waves and THREADS_PER_BLOCK 1024 groups.  `tiles` generalises BLOCK_SIZE (512   */
/* in the reference, Sh/Constants.cginc:3); count = tiles * 1024.  tiles must be a multiple of  */
/* 128 and <= 4096 (BlockSum runs tiles/4 threads in waves of 32, Sh/Sorting/Scan.compute:50).  */
/* Used only to pin orc_sort_pairs.                                                             */
/* ------------------------------------------------------------------------------------------- */
#define THREADS_PER_BLOCK 1024
#define WARP_SIZE 32
#define BUCKET_SIZE 256

/* WavePrefixSum over one 32-lane wave: exclusive prefix of x within [wave_base, wave_base+32) */
static void wave_prefix_sum32(const uint32_t* x, uint32_t* out, uint32_t nthreads)
{
    for (uint32_t w = 0; w < nthreads; w += WARP_SIZE) {
        uint32_t s = 0;
        for (uint32_t l = 0; l < WARP_SIZE && w + l < nthreads; l++) { out[w + l] = s; s += x[w + l]; }
    }
}

/* IntraBlockScan  Sh/Sorting/LocalRadixSort.compute:29-51 */
static void intra_block_scan(const uint32_t* pred, uint32_t* result)
{
    uint32_t warp_result[THREADS_PER_BLOCK];
    uint32_t scan_tile[THREADS_PER_BLOCK / WARP_SIZE];
    uint32_t scan_tile2[THREADS_PER_BLOCK / WARP_SIZE];
    wave_prefix_sum32(pred, warp_result, THREADS_PER_BLOCK);       /* WavePrefixCountBits :33 */
    for (uint32_t t = 0; t < THREADS_PER_BLOCK; t++)
        if (t % WARP_SIZE == WARP_SIZE - 1) scan_tile[t / WARP_SIZE] = warp_result[t] + pred[t]; /* :37-40 */
    wave_prefix_sum32(scan_tile, scan_tile2, WARP_SIZE);            /* :43-47 (threadId < 32) */
    for (uint32_t t = 0; t < THREADS_PER_BLOCK; t++)
        result[t] = warp_result[t] + scan_tile2[t / WARP_SIZE];    /* :50 */
}

int orc_sort_pairs_literal(uint32_t* keys, uint32_t* values, uint32_t tiles)
{
    if (tiles == 0 || tiles % 128 != 0 || tiles > 4096) return -1;
    const size_t count = (size_t)tiles * THREADS_PER_BLOCK;
    uint32_t* blk_keys = (uint32_t*)malloc(count * 4);    /* sortedBlocksKeysData   */
    uint32_t* blk_vals = (uint32_t*)malloc(count * 4);    /* sortedBlocksValuesData */
    uint32_t* offsets = (uint32_t*)malloc((size_t)tiles * BUCKET_SIZE * 4);   /* offsetsData */
    uint32_t* sizes = (uint32_t*)malloc((size_t)tiles * BUCKET_SIZE * 4);     /* sizesData   */
    const uint32_t scan_groups = tiles / (THREADS_PER_BLOCK / BUCKET_SIZE);   /* ComputeBufferSorter.cs:112 */
    uint32_t* block_sums = (uint32_t*)malloc((size_t)scan_groups * 4);        /* blockSumsData */

    for (uint32_t bit_offset = 0; bit_offset < 32; bit_offset += 8) {         /* Sort() :102 */
        /* ---- LocalRadixSort  Sh/Sorting/LocalRadixSort.compute:53-134, `tiles` groups ---- */
        for (uint32_t g = 0; g < tiles; g++) {
            uint32_t sort_tile[THREADS_PER_BLOCK], values_tile[THREADS_PER_BLOCK];
            uint32_t pred[THREADS_PER_BLOCK], true_before[THREADS_PER_BLOCK];
            uint32_t nk[THREADS_PER_BLOCK], nv[THREADS_PER_BLOCK];
            memcpy(sort_tile, keys + (size_t)g * THREADS_PER_BLOCK, sizeof sort_tile);       /* :59 */
            memcpy(values_tile, values + (size_t)g * THREADS_PER_BLOCK, sizeof values_tile); /* :60 */
            for (uint32_t shift = bit_offset; shift < bit_offset + 8; shift++) {             /* :64 */
                for (uint32_t t = 0; t < THREADS_PER_BLOCK; t++) pred[t] = (sort_tile[t] >> shift) & 1u; /* :71 */
                intra_block_scan(pred, true_before);                                         /* :77 */
                const uint32_t false_total = THREADS_PER_BLOCK -
                    (true_before[THREADS_PER_BLOCK - 1] + pred[THREADS_PER_BLOCK - 1]);     /* :81-84 */
                for (uint32_t t = 0; t < THREADS_PER_BLOCK; t++) {
                    uint32_t dst = pred[t] ? true_before[t] + false_total : t - true_before[t]; /* :87-88 */
                    nk[dst] = sort_tile[t];
                    nv[dst] = values_tile[t];
                }
                memcpy(sort_tile, nk, sizeof nk);
                memcpy(values_tile, nv, sizeof nv);
            }
            memcpy(blk_keys + (size_t)g * THREADS_PER_BLOCK, sort_tile, sizeof sort_tile);   /* :99 */
            memcpy(blk_vals + (size_t)g * THREADS_PER_BLOCK, values_tile, sizeof values_tile); /* :100 */

            uint32_t radix_tile[THREADS_PER_BLOCK], offsets_tile[BUCKET_SIZE], sizes_tile[BUCKET_SIZE];
            for (uint32_t t = 0; t < THREADS_PER_BLOCK; t++)
                radix_tile[t] = (sort_tile[t] >> bit_offset) & (BUCKET_SIZE - 1);           /* :102 */
            memset(offsets_tile, 0, sizeof offsets_tile);                                    /* :104-108 */
            memset(sizes_tile, 0, sizeof sizes_tile);
            for (uint32_t t = 1; t < THREADS_PER_BLOCK; t++)
                if (radix_tile[t - 1] != radix_tile[t]) offsets_tile[radix_tile[t]] = t;     /* :111-114 */
            for (uint32_t t = 1; t < THREADS_PER_BLOCK; t++)
                if (radix_tile[t - 1] != radix_tile[t]) {
                    uint32_t r = radix_tile[t - 1];
                    sizes_tile[r] = t - offsets_tile[r];                                     /* :117-121 */
                }
            {
                uint32_t r = radix_tile[THREADS_PER_BLOCK - 1];
                sizes_tile[r] = THREADS_PER_BLOCK - offsets_tile[r];                         /* :122-126 */
            }
            for (uint32_t d = 0; d < BUCKET_SIZE; d++) {
                offsets[(size_t)g * BUCKET_SIZE + d] = offsets_tile[d];                      /* :131 */
                sizes[(size_t)g + (size_t)d * tiles] = sizes_tile[d];                        /* :132 (BLOCK_SIZE -> tiles) */
            }
        }
        /* ---- PreScan  Sh/Sorting/Scan.compute:15-48, scan_groups groups of 1024 ---- */
        for (uint32_t g = 0; g < scan_groups; g++) {
            uint32_t* data = sizes + (size_t)g * THREADS_PER_BLOCK;
            uint32_t wave_prefix[THREADS_PER_BLOCK];
            uint32_t scan_tile[THREADS_PER_BLOCK / WARP_SIZE], warp_prefix[THREADS_PER_BLOCK / WARP_SIZE];
            wave_prefix_sum32(data, wave_prefix, THREADS_PER_BLOCK);                         /* :23 */
            for (uint32_t t = WARP_SIZE - 1; t < THREADS_PER_BLOCK; t += WARP_SIZE)
                scan_tile[t / WARP_SIZE] = wave_prefix[t] + data[t];                         /* :25-28 */
            wave_prefix_sum32(scan_tile, warp_prefix, THREADS_PER_BLOCK / WARP_SIZE);        /* :31-36 */
            block_sums[g] = warp_prefix[THREADS_PER_BLOCK / WARP_SIZE - 1] +
                            scan_tile[THREADS_PER_BLOCK / WARP_SIZE - 1];                    /* :38-41 */
            for (uint32_t t = 0; t < THREADS_PER_BLOCK; t++)
                data[t] = wave_prefix[t] + warp_prefix[t / WARP_SIZE];                       /* :45 */
        }
        /* ---- BlockSum  Sh/Sorting/Scan.compute:50-84, one group of scan_groups threads ---- */
        {
            uint32_t* wave_prefix = (uint32_t*)malloc((size_t)scan_groups * 4);
            uint32_t scan_tile[32], warp_prefix[32];
            const uint32_t nwaves = scan_groups / WARP_SIZE;
            wave_prefix_sum32(block_sums, wave_prefix, scan_groups);                         /* :64 */
            for (uint32_t w = 0; w < nwaves; w++)
                scan_tile[w] = wave_prefix[w * WARP_SIZE + WARP_SIZE - 1] +
                               block_sums[w * WARP_SIZE + WARP_SIZE - 1];                    /* :66-69 */
            wave_prefix_sum32(scan_tile, warp_prefix, nwaves);                               /* :72-78 */
            for (uint32_t t = 0; t < scan_groups; t++)
                block_sums[t] = wave_prefix[t] + warp_prefix[t / WARP_SIZE];                 /* :81 */
            free(wave_prefix);
        }
        /* ---- GlobalScan  Sh/Sorting/Scan.compute:86-96 ---- */
        for (uint32_t g = 0; g < scan_groups; g++)
            for (uint32_t t = 0; t < THREADS_PER_BLOCK; t++)
                sizes[(size_t)g * THREADS_PER_BLOCK + t] += block_sums[g];                   /* :93-95 */
        /* ---- GlobalRadixSort  Sh/Sorting/GlobalRadixSort.compute:20-40 ---- */
        for (uint32_t g = 0; g < tiles; g++)
            for (uint32_t t = 0; t < THREADS_PER_BLOCK; t++) {
                const uint32_t key = blk_keys[(size_t)g * THREADS_PER_BLOCK + t];            /* :26 */
                const uint32_t value = blk_vals[(size_t)g * THREADS_PER_BLOCK + t];          /* :27 */
                const uint32_t radix = (key >> bit_offset) & (BUCKET_SIZE - 1);              /* :35 */
                const uint32_t index_output = sizes[(size_t)g + (size_t)radix * tiles] + t -
                                              offsets[(size_t)g * BUCKET_SIZE + radix];      /* :36 */
                if (index_output >= count) { /* D3D would drop the write; flag it instead */
                    free(blk_keys); free(blk_vals); free(offsets); free(sizes); free(block_sums);
                    return -2;
                }
                keys[index_output] = key;                                                    /* :38 */
                values[index_output] = value;                                                /* :39 */
            }
    }
    free(blk_keys); free(blk_vals); free(offsets); free(sizes); free(block_sums);
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a-6  DistributeKeys     Sc/MeshBufferContainer.cs:154-169                                    */
/* ------------------------------------------------------------------------------------------- */
void orc_distribute_keys(uint32_t* keys, uint32_t n)
{
    if (n == 0) return;
    uint32_t new_current = 0;                    /* :158 */
    uint32_t old_current = keys[0];              /* :159 */
    keys[0] = new_current;                       /* :160 */
    for (uint32_t i = 1; i < n; i++) {           /* :161 */
        uint32_t diff = keys[i] - old_current;   /* uint arithmetic */
        new_current += diff > 1u ? diff : 1u;    /* Math.Max(diff, 1)  :163 */
        old_current = keys[i];                   /* :164 */
        keys[i] = new_current;                   /* :165 */
    }
}

/* OpenMP form (CPU baseline only): new[i] = sum_{j=1..i} max(old[j] - old[j-1], 1), a two-level prefix sum. */
void orc_distribute_keys_mt(uint32_t* keys, uint32_t n, int threads)
{
#ifdef _OPENMP
    if (threads > 1 && n >= 65536u) {
        uint32_t* part = (uint32_t*)calloc((size_t)threads + 1, 4);
        uint32_t* first_old = (uint32_t*)calloc((size_t)threads + 1, 4);
        const size_t chunk = ((size_t)n + (size_t)threads - 1) / (size_t)threads;
        for (int t = 0; t < threads; t++) {           /* the old key just before each chunk, before anything is overwritten */
            const size_t lo = (size_t)t * chunk;
            first_old[t] = (lo > 0 && lo - 1 < n) ? keys[lo - 1] : 0u;
        }
#pragma omp parallel num_threads(threads)
        {
            /* chunks dealt to the threads actually granted, as in orc_sort_pairs_mt */
            const int nth = omp_get_num_threads();
            for (int c = omp_get_thread_num(); c < threads; c += nth) {
                const size_t lo = (size_t)c * chunk < n ? (size_t)c * chunk : n;
                const size_t hi = lo + chunk < n ? lo + chunk : n;
                uint32_t sum = 0, prev = first_old[c];
                for (size_t i = lo; i < hi; i++) {
                    if (i > 0) { const uint32_t diff = keys[i] - prev; sum += diff > 1u ? diff : 1u; }
                    prev = keys[i];
                }
                part[c + 1] = sum;
            }
#pragma omp barrier
#pragma omp single
            for (int u = 0; u < threads; u++) part[u + 1] += part[u];
            for (int c = omp_get_thread_num(); c < threads; c += nth) {
                const size_t lo = (size_t)c * chunk < n ? (size_t)c * chunk : n;
                const size_t hi = lo + chunk < n ? lo + chunk : n;
                uint32_t run = part[c], prev = first_old[c];
                for (size_t i = lo; i < hi; i++) {
                    const uint32_t old = keys[i];
                    if (i > 0) { const uint32_t diff = old - prev; run += diff > 1u ? diff : 1u; }
                    prev = old;
                    keys[i] = run;
                }
            }
        }
        free(part); free(first_old);
        return;
    }
#endif
    (void)threads;
    orc_distribute_keys(keys, n);
}

/* ------------------------------------------------------------------------------------------- */
/* a-7  TreeConstructor    Sh/BVH/BVH.compute:18-149                                            */
/* ------------------------------------------------------------------------------------------- */

/* :18-21   31 - firstbithigh(v);  firstbithigh(0) = -1  =>  32 */
static inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }

/* :23-33 */
static inline int delta(const uint32_t* codes, int64_t x, int64_t y, int num_objects)
{
    if (x >= 0 && x <= num_objects - 1 && y >= 0 && y <= num_objects - 1)
        return clz32(codes[x] ^ codes[y]);
    return -1;
}

static inline int sign_i(int v) { return (v > 0) - (v < 0); }

/* :35-52.  HLSL evaluates idx + lmax*d in 32-bit two's complement (uint*int); int64 here is the
 * same value wherever the reference's result is in range, and out of range both give delta -1. */
static void determine_range(const uint32_t* codes, int num_objects, int idx, int* first, int* last)
{
    const int d = sign_i(delta(codes, idx, (int64_t)idx + 1, num_objects) -
                         delta(codes, idx, (int64_t)idx - 1, num_objects));       /* :37 */
    const int dmin = delta(codes, idx, (int64_t)idx - d, num_objects);            /* :38 */
    uint32_t lmax = 2;                                                             /* :39 */
    while (delta(codes, idx, (int64_t)idx + (int64_t)(int32_t)(lmax * (uint32_t)d), num_objects) > dmin) /* :40 */
        lmax = lmax * 2;                                                           /* :41 */
    int l = 0;                                                                     /* :42 */
    for (uint32_t t = lmax / 2; t >= 1; t /= 2) {                                  /* :43 */
        int64_t probe = (int64_t)idx + (int64_t)(int32_t)(((uint32_t)l + t) * (uint32_t)d);
        if (delta(codes, idx, probe, num_objects) > dmin) l += (int)t;             /* :45-46 */
    }
    const int j = idx + l * d;                                                     /* :49 */
    *first = idx < j ? idx : j;                                                    /* :50 */
    *last = idx > j ? idx : j;
}

/* :54-92 */
static int find_split(const uint32_t* codes, int first, int last)
{
    const uint32_t first_code = codes[first];
    const uint32_t last_code = codes[last];
    if (first_code == last_code) return (first + last) >> 1;      /* :61-62 */
    const int common_prefix = clz32(first_code ^ last_code);      /* :67 */
    int split = first;                                            /* :73 */
    int step = last - first;                                      /* :74 */
    do {
        step = (step + 1) >> 1;                                   /* :78 */
        const int new_split = split + step;                       /* :79 */
        if (new_split < last) {                                   /* :81 */
            const uint32_t split_code = codes[new_split];
            const int split_prefix = clz32(first_code ^ split_code);
            if (split_prefix > common_prefix) split = new_split;  /* :85-86 */
        }
    } while (step > 1);                                           /* :89 */
    return split;
}

int orc_build_tree(uint32_t n, const uint32_t* sorted_keys, lbvh_internal_node* internal,
                   lbvh_leaf_node* leaf, int threads)
{
    (void)threads;
    if (n < 2) return -1;               /* threadId < n - 1 underflows in the reference (:101) */
    int bad = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static) reduction(| : bad)
#endif
    for (int64_t tid = 0; tid < (int64_t)n - 1; tid++) {           /* :101 */
        const uint32_t thread_id = (uint32_t)tid;
        int first, last;
        determine_range(sorted_keys, (int)n, (int)thread_id, &first, &last);     /* :103 */
        const int split = find_split(sorted_keys, first, last);                  /* :109 */
        if (split < 0 || (uint32_t)split + 1 >= n) { bad |= 1; continue; }
        internal[thread_id].index = thread_id;                                   /* :111 */
        if (split == first) {                                                    /* :114 */
            leaf[split].parent = thread_id;                                      /* :116-120 */
            leaf[split].index = (uint32_t)split;
            internal[thread_id].leftNode = (uint32_t)split;                      /* :121 */
            internal[thread_id].leftNodeType = LBVH_LEAF_NODE;                   /* :122 */
        } else {
            internal[split].parent = thread_id;                                  /* :126 */
            internal[thread_id].leftNode = (uint32_t)split;                      /* :127 */
            internal[thread_id].leftNodeType = LBVH_INTERNAL_NODE;               /* :128 */
        }
        if (split + 1 == last) {                                                 /* :132 */
            leaf[split + 1].parent = thread_id;                                  /* :134-138 */
            leaf[split + 1].index = (uint32_t)split + 1;
            internal[thread_id].rightNode = (uint32_t)split + 1;                 /* :139 */
            internal[thread_id].rightNodeType = LBVH_LEAF_NODE;                  /* :140 */
        } else {
            internal[split + 1].parent = thread_id;                              /* :144 */
            internal[thread_id].rightNode = (uint32_t)split + 1;                 /* :145 */
            internal[thread_id].rightNodeType = LBVH_INTERNAL_NODE;              /* :146 */
        }
    }
    return bad ? -2 : 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a-8  BVHConstructor (refit)     Sh/BVH/BVH.compute:152-220                                   */
/* Threads run one after another here; min/max are exact, so every interleaving the GPU can     */
/* produce gives the same boxes.                                                                */
/* ------------------------------------------------------------------------------------------- */

/* :152-170.  HLSL min/max return the non-NaN operand, like fminf/fmaxf. */
static lbvh_aabb merge_aabb(lbvh_aabb l, lbvh_aabb r)
{
    lbvh_aabb o;
    for (int k = 0; k < 3; k++) {
        o.min[k] = fminf(l.min[k], r.min[k]);
        o.max[k] = fmaxf(l.max[k], r.max[k]);
    }
    o._dummy0 = 0.0f;
    o._dummy1 = 0.0f;
    return o;
}

int orc_refit(uint32_t n, const lbvh_internal_node* internal, const lbvh_leaf_node* leaf,
              const lbvh_aabb* triangle_aabb, const uint32_t* sorted_indices, lbvh_aabb* bvh)
{
    if (n < 2) return -1;
    uint32_t* atomics = (uint32_t*)calloc(n, 4);       /* DataBuffer<uint>(.., 0)  Sc/BVHConstructor.cs:41 */
    int rc = 0;
    for (uint32_t tid = 0; tid < n; tid++) {           /* :179 */
        uint32_t parent = leaf[tid].parent;            /* :181 */
        uint32_t guard = 0;
        while (parent != 0xFFFFFFFFu) {                /* :182 */
            if (parent >= n - 1 || ++guard > 64) { rc = -2; break; }
            uint32_t old = atomics[parent];            /* InterlockedCompareExchange(.., 0, 1, old)  :185 */
            if (old == 0) atomics[parent] = 1;
            if (old == 0) break;                       /* :186-189 */
            const lbvh_internal_node nd = internal[parent];
            lbvh_aabb lb = nd.leftNodeType == LBVH_INTERNAL_NODE
                               ? bvh[nd.leftNode]
                               : triangle_aabb[sorted_indices[nd.leftNode]];      /* :196-204 */
            lbvh_aabb rb = nd.rightNodeType == LBVH_INTERNAL_NODE
                               ? bvh[nd.rightNode]
                               : triangle_aabb[sorted_indices[nd.rightNode]];     /* :205-213 */
            bvh[parent] = merge_aabb(lb, rb);          /* :215 */
            parent = nd.parent;                        /* :217 */
        }
    }
    free(atomics);
    return rc;
}

/* OpenMP form (CPU baseline only): the reference's own scheme — one walker per leaf, the second arrival at a node
 * merges — with the hand-off made explicit: an acq_rel exchange on the node's flag publishes the first arrival's box
 * to the second (the reference has no such fence). */
int orc_refit_mt(uint32_t n, const lbvh_internal_node* internal, const lbvh_leaf_node* leaf,
                 const lbvh_aabb* triangle_aabb, const uint32_t* sorted_indices, lbvh_aabb* bvh, int threads)
{
#ifdef _OPENMP
    if (threads > 1 && n >= 65536u) {
        uint32_t* atomics = (uint32_t*)calloc(n, 4);
        int rc = 0;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(| : rc)
        for (int64_t tid = 0; tid < (int64_t)n; tid++) {
            uint32_t parent = leaf[tid].parent;
            uint32_t guard = 0;
            while (parent != 0xFFFFFFFFu) {
                if (parent >= n - 1 || ++guard > 64) { rc |= 2; break; }
                if (__atomic_exchange_n(&atomics[parent], 1u, __ATOMIC_ACQ_REL) == 0) break;
                const lbvh_internal_node nd = internal[parent];
                const lbvh_aabb lb = nd.leftNodeType == LBVH_INTERNAL_NODE ? bvh[nd.leftNode] : triangle_aabb[sorted_indices[nd.leftNode]];
                const lbvh_aabb rb = nd.rightNodeType == LBVH_INTERNAL_NODE ? bvh[nd.rightNode] : triangle_aabb[sorted_indices[nd.rightNode]];
                bvh[parent] = merge_aabb(lb, rb);
                parent = nd.parent;
            }
        }
        free(atomics);
        return rc ? -2 : 0;
    }
#endif
    (void)threads;
    return orc_refit(n, internal, leaf, triangle_aabb, sorted_indices, bvh);
}

/* ------------------------------------------------------------------------------------------- */
/* a-9  Raytracing kernel, ray gen + traversal    Sh/Raytracing/Raytracing.compute:23-176       */
/* ------------------------------------------------------------------------------------------- */

typedef struct { float origin[3], dir[3], inv_dir[3]; } ray_t;      /* :23-28 */

/* RayBoxIntersection  :75-87.  HLSL min/max = fminf/fmaxf (non-NaN operand wins). */
static int ray_box_entry(const float bmin[3], const float bmax[3], const float origin[3],
                         const float inv_dir[3], float* entry)
{
    float tmin1[3], tmax1[3];
    for (int k = 0; k < 3; k++) {
        float t1 = (bmin[k] - origin[k]) * inv_dir[k];
        float t2 = (bmax[k] - origin[k]) * inv_dir[k];
        tmin1[k] = fminf(t1, t2);
        tmax1[k] = fmaxf(t1, t2);
    }
    const float tmin = fmaxf(tmin1[0], fmaxf(tmin1[1], tmin1[2]));
    const float tmax = fminf(tmax1[0], fminf(tmax1[1], tmax1[2]));
    *entry = tmin;                 /* the distance at which the slab test says the ray enters the box */
    return tmax > tmin && tmax > 0.0f;
}

int orc_ray_box(const float bmin[3], const float bmax[3], const float origin[3],
                const float inv_dir[3])
{
    float entry;
    return ray_box_entry(bmin, bmax, origin, inv_dir, &entry);
}

static inline void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline float dot3(const float a[3], const float b[3])
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

/* RayTriangleIntersection  :37-73.  Returns distance (LBVH_MAX_FLOAT = miss). */
static float ray_triangle(const float orig[3], const float dir[3], const float v0[3],
                          const float v1[3], const float v2[3], float* out_u, float* out_v)
{
    float e1[3], e2[3], pvec[3], tvec[3], qvec[3];
    for (int k = 0; k < 3; k++) { e1[k] = v1[k] - v0[k]; e2[k] = v2[k] - v0[k]; }
    cross3(dir, e2, pvec);
    const float det = dot3(e1, pvec);
    if (det < 1e-8f && det > -1e-8f) return LBVH_MAX_FLOAT;           /* :47-51 */
    const float inv_det = 1.0f / det;
    for (int k = 0; k < 3; k++) tvec[k] = orig[k] - v0[k];
    const float u = dot3(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return LBVH_MAX_FLOAT;                  /* :56-60 */
    cross3(tvec, e1, qvec);
    const float v = dot3(dir, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return LBVH_MAX_FLOAT;              /* :64-68 */
    *out_u = u;
    *out_v = v;
    return dot3(e2, qvec) * inv_det;                                  /* :70 — no t > 0 test */
}

/* ray generation  :108-126 */
void orc_make_ray(const lbvh_camera* cam, uint32_t px, uint32_t py, float origin[3], float dir[3],
                  float inv_dir[3])
{
    const float near = cam->near_plane;
    const float fov = cam->camera_fov;
    const float height = 2.0f * near * fov;                                          /* :110 */
    const float width = (float)cam->screen_width * height / (float)cam->screen_height; /* :111 */
    float d[3];
    d[0] = -width / 2.0f + width / (float)cam->screen_width * ((float)px + 0.5f);    /* :115 */
    d[1] = -height / 2.0f + height / (float)cam->screen_height * ((float)py + 0.5f); /* :116 */
    d[2] = -near;                                                                    /* :117 */
    const float* m = cam->camera_to_world;
    /* mul(M, float4(0,0,0,1)).xyz and mul(M, float4(dir,0)).xyz  :120-121; row . vector,
     * summed left to right */
    float w[3];
    for (int r = 0; r < 3; r++) {
        origin[r] = ((m[4 * r + 0] * 0.0f + m[4 * r + 1] * 0.0f) + m[4 * r + 2] * 0.0f) + m[4 * r + 3] * 1.0f;
        w[r] = ((m[4 * r + 0] * d[0] + m[4 * r + 1] * d[1]) + m[4 * r + 2] * d[2]) + m[4 * r + 3] * 0.0f;
    }
    const float len = sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);               /* normalize :125 */
    for (int r = 0; r < 3; r++) {
        dir[r] = w[r] / len;
        inv_dir[r] = 1.0f / dir[r];                                                   /* :126 */
    }
}

/* CheckTriangle  :89-103 */
/* the triangle test on its own (tests: which triangles does a ray hit at exactly which t) */
float orc_ray_triangle(const float orig[3], const float dir[3], const float a[3], const float b[3], const float c[3])
{
    float u, v;
    return ray_triangle(orig, dir, a, b, c, &u, &v);
}

/* `fast_rule` (NOT the reference: the semantics of the library's LBVH_TRACE_FAST / _FAST_EXACT): a computed t that
 * lies before the distance at which the ray enters the triangle's own box does not count.  The reference, which prunes nothing,
 * reports such a t (fp32 noise of the triangle test on a ray almost inside the triangle's plane); a walk that skips boxes
 * entered beyond its best hit can only be order-independent — and equal to a CPU restatement — without them.  With
 * fast_rule = 0 this is CheckTriangle as written. */
static void check_triangle(uint32_t triangle_index, const ray_t* ray, const lbvh_scene* s,
                           lbvh_hit* result, lbvh_trace_stats* st, int fast_rule)
{
    st->leaf_tests++;
    const lbvh_aabb* b = &s->triangle_aabb[triangle_index];
    float entry;
    if (ray_box_entry(b->min, b->max, ray->origin, ray->inv_dir, &entry)) {           /* :91 */
        st->tri_tests++;
        const lbvh_triangle* t = &s->triangles[triangle_index];                       /* :93 */
        float u = 0.0f, v = 0.0f;
        const float dist = ray_triangle(ray->origin, ray->dir, t->a, t->b, t->c, &u, &v);
        if (fast_rule && dist < entry) return;
        if (dist < result->t) {                                                        /* :95 strict */
            result->t = dist;
            result->tri = triangle_index;                                              /* :97 */
            result->u = u;
            result->v = v;
        }
    }
}

/* the traversal loop  :128-176 for one ray */
static int trace_one(const lbvh_scene* s, const ray_t* ray, lbvh_hit* result, lbvh_trace_stats* st, int fast_rule)
{
    result->t = LBVH_MAX_FLOAT;                 /* :129 */
    result->tri = 0;                            /* :130 */
    result->u = 0.0f;                           /* :131 */
    result->v = 0.0f;
    uint32_t stack[64];                         /* :133 */
    uint32_t sp = 0;
    stack[sp] = 0;                              /* :135 */
    sp = 1;
    int overflow = 0;
    while (sp != 0) {                           /* :138 */
        sp--;
        const uint32_t index = stack[sp];       /* :141 */
        st->pops++;
        const lbvh_aabb* nb = &s->bvh[index];
        if (!orc_ray_box(nb->min, nb->max, ray->origin, ray->inv_dir)) continue;    /* :143-146 */
        st->box_hits++;
        const lbvh_internal_node* nd = &s->internal_nodes[index];
        if (nd->leftNodeType == LBVH_INTERNAL_NODE) {                                /* :151 */
            if (sp >= 64) { overflow = 1; break; }
            stack[sp++] = nd->leftNode;                                              /* :153-154 */
        } else {
            const uint32_t tri = s->sorted_indices[s->leaf_nodes[nd->leftNode].index]; /* :158 */
            check_triangle(tri, ray, s, result, st, fast_rule);                                 /* :159 */
        }
        if (nd->rightNodeType == LBVH_INTERNAL_NODE) {                               /* :166 */
            if (sp >= 64) { overflow = 1; break; }
            stack[sp++] = nd->rightNode;                                             /* :168-169 */
        } else {
            const uint32_t tri = s->sorted_indices[s->leaf_nodes[nd->rightNode].index]; /* :173 */
            check_triangle(tri, ray, s, result, st, fast_rule);                                 /* :174 */
        }
    }
    if (result->t < LBVH_MAX_FLOAT) st->hits++;                                      /* :184 alpha */
    return overflow;
}

int orc_trace_primary(const lbvh_camera* cam, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                      int32_t x_step, int32_t y_step, const lbvh_scene* scene, lbvh_hit* hits,
                      lbvh_trace_stats* stats, int threads)
{
    return orc_trace_primary_rule(cam, x0, y0, x1, y1, x_step, y_step, scene, hits, stats, threads, 0);
}

/* fast_rule = 0: the reference.  fast_rule = 1: the reference's loop with the fast modes' accept rule (check_triangle) — what
 * LBVH_TRACE_FAST_EXACT must return word for word and LBVH_TRACE_FAST in t; equal to the reference's frame wherever the
 * reference's winner is not a t in front of its own triangle's box. */
int orc_trace_primary_rule(const lbvh_camera* cam, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                           int32_t x_step, int32_t y_step, const lbvh_scene* scene, lbvh_hit* hits,
                           lbvh_trace_stats* stats, int threads, int fast_rule)
{
    (void)threads;
    if (x_step < 1 || y_step < 1 || x1 < x0 || y1 < y0) return -1;
    const int64_t w = (x1 - x0 + x_step - 1) / x_step;
    const int64_t h = (y1 - y0 + y_step - 1) / y_step;
    uint64_t pops = 0, box_hits = 0, leaf_tests = 0, tri_tests = 0, nhits = 0;
    int overflow = 0;
    /* 8x8-sample tiles handed out dynamically: neighbouring rays walk the same nodes (cache), and thousands of
     * tiles keep every core busy to the end (whole rows left 256 threads with 270 chunks of very unequal cost) */
    const int64_t tw = (w + 7) / 8, th = (h + 7) / 8;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1) \
    reduction(+ : pops, box_hits, leaf_tests, tri_tests, nhits) reduction(| : overflow)
#endif
    for (int64_t tile = 0; tile < tw * th; tile++) {
        const int64_t j0 = (tile / tw) * 8, i0 = (tile % tw) * 8;
        for (int64_t j = j0; j < j0 + 8 && j < h; j++) {
            for (int64_t i = i0; i < i0 + 8 && i < w; i++) {
                ray_t ray;
                orc_make_ray(cam, (uint32_t)(x0 + i * x_step), (uint32_t)(y0 + j * y_step), ray.origin,
                             ray.dir, ray.inv_dir);
                lbvh_trace_stats st = {0, 0, 0, 0, 0};
                overflow |= trace_one(scene, &ray, &hits[j * w + i], &st, fast_rule);
                pops += st.pops; box_hits += st.box_hits; leaf_tests += st.leaf_tests;
                tri_tests += st.tri_tests; nhits += st.hits;
            }
        }
    }
    if (stats) {
        stats->pops = pops; stats->box_hits = box_hits; stats->leaf_tests = leaf_tests;
        stats->tri_tests = tri_tests; stats->hits = nhits;
    }
    return overflow ? -3 : 0;
}

/* ------------------------------------------------------------------------------------------- */
/* a-9 tail  shading     Sh/Raytracing/Raytracing.compute:178-184                               */
/* ------------------------------------------------------------------------------------------- */

/* float -> IEEE half, round to nearest even (the RGBA16F store of the render target) */
static uint16_t float_to_half(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));   /* nan / inf */
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                  /* rounds to inf */
    if (x < 0x33000001u) return (uint16_t)sign;                                                /* rounds to 0 */
    uint32_t e = x >> 23, m = x & 0x7FFFFFu;
    if (e < 113) {                                                                             /* subnormal half */
        m |= 0x800000u;
        const uint32_t shift = 126 - e;              /* 14 .. 24 */
        uint32_t h = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((e - 112) << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

static void texel_rgba(const uint8_t* tex, int w, int x, int y, float c[4])
{
    const uint8_t* p = tex + ((size_t)y * w + x) * 4;
    for (int k = 0; k < 4; k++) c[k] = (float)p[k] / 255.0f;
}

/* SampleLevel(linearClampSampler, uv, 0)  :183 — bilinear on texel centres, clamp addressing, fp32 */
static void sample_bilinear_clamp(const uint8_t* tex, int w, int h, float u, float v, float out[4])
{
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    const float xc0 = fminf(fmaxf(xf, 0.0f), (float)(w - 1)), xc1 = fminf(fmaxf(xf + 1.0f, 0.0f), (float)(w - 1));
    const float yc0 = fminf(fmaxf(yf, 0.0f), (float)(h - 1)), yc1 = fminf(fmaxf(yf + 1.0f, 0.0f), (float)(h - 1));
    float c00[4], c10[4], c01[4], c11[4];
    texel_rgba(tex, w, (int)xc0, (int)yc0, c00);
    texel_rgba(tex, w, (int)xc1, (int)yc0, c10);
    texel_rgba(tex, w, (int)xc0, (int)yc1, c01);
    texel_rgba(tex, w, (int)xc1, (int)yc1, c11);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    for (int k = 0; k < 4; k++) out[k] = (c00[k] * gx + c10[k] * fx) * gy + (c01[k] * gx + c11[k] * fx) * fy;
}

void orc_shade(const lbvh_hit* hits, size_t count, const lbvh_triangle* tris, const uint8_t* tex, int32_t tex_w,
               int32_t tex_h, uint16_t* rgba16f)
{
    for (size_t i = 0; i < count; i++) {
        const lbvh_hit* r = &hits[i];
        const lbvh_triangle* t = &tris[r->tri];                                              /* :178 */
        const float w = (1.0f - r->u) - r->v;
        const float tu = (w * t->a_uv[0] + r->u * t->b_uv[0]) + r->v * t->c_uv[0];          /* :179 */
        const float tv = (w * t->a_uv[1] + r->u * t->b_uv[1]) + r->v * t->c_uv[1];
        float n[3];
        for (int k = 0; k < 3; k++) n[k] = (w * t->a_normal[k] + r->u * t->b_normal[k]) + r->v * t->c_normal[k]; /* :180 */
        const float light_dir = 0.57735026f;               /* const float lightDir = normalize(float3(1,1,1))  :181 */
        const float lambert = fmaxf(0.4f, (light_dir * n[0] + light_dir * n[1]) + light_dir * n[2]);
        float c[4];
        sample_bilinear_clamp(tex, tex_w, tex_h, tu, tv, c);                                 /* :183 */
        rgba16f[4 * i + 0] = float_to_half(c[0] * lambert);
        rgba16f[4 * i + 1] = float_to_half(c[1] * lambert);
        rgba16f[4 * i + 2] = float_to_half(c[2] * lambert);
        rgba16f[4 * i + 3] = float_to_half(r->t != LBVH_MAX_FLOAT ? 1.0f : 0.0f);           /* :184 */
    }
}

/* IEEE half -> float (exact) */
static float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else {                                               /* subnormal half: normalise */
            int shift = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; shift++; }
            bits = sign | ((uint32_t)(113 - shift) << 23) | ((mm & 0x3FFu) << 13);
        }
    } else if (e == 31) bits = sign | 0x7F800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* Hidden/ImageComposer, Assets/_Shaders/ImageComposer.shader:44-52: ret = lerp(col.rgb, colObject.rgb, colObject.a),
 * alpha 1; lerp(a, b, t) = a + t * (b - a), each operation rounded to fp32; RGBA16F in and out. */
void orc_compose(const uint16_t* background, const uint16_t* object, size_t count, uint16_t* out)
{
    for (size_t i = 0; i < count; i++) {
        const float a = half_to_float(object[4 * i + 3]);
        for (int k = 0; k < 3; k++) {
            const float bg = half_to_float(background[4 * i + k]), ob = half_to_float(object[4 * i + k]);
            const float d = ob - bg;
            const float t = a * d;
            out[4 * i + k] = float_to_half(bg + t);
        }
        out[4 * i + 3] = float_to_half(1.0f);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* dynamic scene + secondary rays.  NOT in the reference (it has neither); these  */
/* restate include/lbvh.h's definitions so the GPU kernels have a bit-exact checker.                  */
/* ------------------------------------------------------------------------------------------- */

void orc_animate(const lbvh_triangle* rest, uint32_t n, const uint32_t* body, const float* centres, float c, float s,
                 lbvh_triangle* out)
{
    for (uint32_t i = 0; i < n; i++) {
        const float* ctr = &centres[4 * body[i]];
        lbvh_triangle t = rest[i];
        float* pos[3] = {t.a, t.b, t.c};
        float* nrm[3] = {t.a_normal, t.b_normal, t.c_normal};
        for (int k = 0; k < 3; k++) {
            const float x = pos[k][0] - ctr[0], z = pos[k][2] - ctr[2];
            pos[k][0] = (c * x + s * z) + ctr[0];          /* rotation about Y through the body centre */
            pos[k][2] = (c * z - s * x) + ctr[2];
            const float nx = nrm[k][0], nz = nrm[k][2];
            nrm[k][0] = c * nx + s * nz;
            nrm[k][2] = c * nz - s * nx;
        }
        out[i] = t;
    }
}

void orc_path_begin(const lbvh_camera* cam, lbvh_path_state* states)
{
    for (int32_t y = 0; y < cam->screen_height; y++)
        for (int32_t x = 0; x < cam->screen_width; x++) {
            lbvh_path_state* st = &states[(size_t)y * cam->screen_width + x];
            float inv[3];
            orc_make_ray(cam, (uint32_t)x, (uint32_t)y, st->origin, st->dir, inv);
            st->alive = 1; st->pad0 = 0.0f; st->pad1 = 0.0f; st->alpha = 0.0f;
            for (int k = 0; k < 3; k++) { st->throughput[k] = 1.0f; st->radiance[k] = 0.0f; }
        }
}

/* closest hit of arbitrary rays over the reference arrays (test infrastructure for the cfg5 extension, which has no reference
 * counterpart: include/lbvh.h lbvh_trace_rays).  Accept rule: t > t_min, strictly nearer — and among triangles hit at EXACTLY the
 * same t the lowest triangle index, whatever order the walk meets them in (the rule of LBVH_TRACE_FAST, which the GPU's per-ray
 * walkers share: a result that does not depend on the visit order); and, for the same reason, a computed t in front of its own
 * triangle's box does not count (check_triangle's fast_rule).  The reference's own first-met rule lives in orc_trace_primary. */
int orc_trace_rays(const lbvh_path_state* states, size_t count, float t_min, const lbvh_scene* s, lbvh_hit* hits,
                   int threads)
{
    (void)threads;
    int overflow = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 256) reduction(| : overflow)
#endif
    for (int64_t ii = 0; ii < (int64_t)count; ii++) {
        lbvh_hit* result = &hits[ii];
        result->t = LBVH_MAX_FLOAT; result->tri = 0; result->u = 0.0f; result->v = 0.0f;
        if (!states[ii].alive) continue;
        ray_t ray;
        for (int k = 0; k < 3; k++) {
            ray.origin[k] = states[ii].origin[k];
            ray.dir[k] = states[ii].dir[k];
            ray.inv_dir[k] = 1.0f / states[ii].dir[k];
        }
        uint32_t stack[64];
        uint32_t sp = 1;
        stack[0] = 0;
        while (sp != 0) {
            const uint32_t index = stack[--sp];
            const lbvh_aabb* nb = &s->bvh[index];
            if (!orc_ray_box(nb->min, nb->max, ray.origin, ray.inv_dir)) continue;
            const lbvh_internal_node* nd = &s->internal_nodes[index];
            const uint32_t child[2] = {nd->leftNode, nd->rightNode}, type[2] = {nd->leftNodeType, nd->rightNodeType};
            for (int side = 0; side < 2; side++) {
                if (type[side] == LBVH_INTERNAL_NODE) {
                    if (sp >= 64) { overflow = 1; break; }
                    stack[sp++] = child[side];
                } else {
                    const uint32_t tri = s->sorted_indices[s->leaf_nodes[child[side]].index];
                    const lbvh_aabb* b = &s->triangle_aabb[tri];
                    float entry;
                    if (!ray_box_entry(b->min, b->max, ray.origin, ray.inv_dir, &entry)) continue;
                    const lbvh_triangle* t = &s->triangles[tri];
                    float u = 0.0f, v = 0.0f;
                    const float dist = ray_triangle(ray.origin, ray.dir, t->a, t->b, t->c, &u, &v);
                    if (dist < entry) continue;          /* the fast modes' accept rule (check_triangle): not in front of its own box */
                    if (dist > t_min && (dist < result->t || (dist == result->t && tri < result->tri))) {
                        result->t = dist; result->tri = tri; result->u = u; result->v = v;
                    }
                }
            }
        }
    }
    return overflow ? -3 : 0;
}

static uint32_t pcg_hash(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

static float path_rnd(uint32_t seed, uint32_t index, uint32_t bounce, uint32_t draw)
{
    const uint32_t h = pcg_hash(pcg_hash(pcg_hash(seed + 0x9E3779B9u * index) + bounce) + draw);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

void orc_path_scatter(const lbvh_scene* s, const lbvh_hit* hits, size_t count, uint32_t bounce, uint32_t seed,
                      float albedo, lbvh_path_state* states)
{
    for (size_t i = 0; i < count; i++) {
        lbvh_path_state* st = &states[i];
        if (!st->alive) continue;
        const lbvh_hit* h = &hits[i];
        if (!(h->t < LBVH_MAX_FLOAT)) {
            const float sk = 0.5f * (st->dir[1] + 1.0f);
            const float sky[3] = {(1.0f - sk) * 1.0f + sk * 0.5f, (1.0f - sk) * 1.0f + sk * 0.7f, (1.0f - sk) * 1.0f + sk * 1.0f};
            for (int k = 0; k < 3; k++) st->radiance[k] = st->radiance[k] + st->throughput[k] * sky[k];
            st->alive = 0;
            continue;
        }
        if (bounce == 0) st->alpha = 1.0f;
        const lbvh_triangle* t = &s->triangles[h->tri];
        float e1[3], e2[3], n[3];
        for (int k = 0; k < 3; k++) { e1[k] = t->b[k] - t->a[k]; e2[k] = t->c[k] - t->a[k]; }
        cross3(e1, e2, n);
        const float nl = sqrtf(dot3(n, n));
        if (nl > 0.0f) { n[0] = n[0] / nl; n[1] = n[1] / nl; n[2] = n[2] / nl; } else { n[0] = 0.0f; n[1] = 1.0f; n[2] = 0.0f; }
        if (dot3(n, st->dir) > 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        for (int k = 0; k < 3; k++) {
            st->origin[k] = st->origin[k] + st->dir[k] * h->t;
            st->throughput[k] = st->throughput[k] * albedo;
        }
        /* uniform point on the unit sphere, Marsaglia 1972, at most 8 tries */
        float p[3] = {0.0f, 0.0f, 1.0f};
        for (uint32_t k = 0; k < 16; k += 2) {
            const float x1 = 2.0f * path_rnd(seed, (uint32_t)i, bounce, k) - 1.0f;
            const float x2 = 2.0f * path_rnd(seed, (uint32_t)i, bounce, k + 1) - 1.0f;
            const float ss = x1 * x1 + x2 * x2;
            if (ss < 1.0f) {
                const float r = sqrtf(1.0f - ss);
                p[0] = 2.0f * x1 * r; p[1] = 2.0f * x2 * r; p[2] = 1.0f - 2.0f * ss;
                break;
            }
        }
        float d[3] = {n[0] + p[0], n[1] + p[1], n[2] + p[2]};
        const float dl = sqrtf(dot3(d, d));
        if (dl > 1e-6f) { d[0] = d[0] / dl; d[1] = d[1] / dl; d[2] = d[2] / dl; } else { d[0] = n[0]; d[1] = n[1]; d[2] = n[2]; }
        for (int k = 0; k < 3; k++) st->dir[k] = d[k];
    }
}

void orc_path_resolve(const lbvh_path_state* states, size_t count, uint16_t* rgba16f)
{
    for (size_t i = 0; i < count; i++) {
        for (int k = 0; k < 3; k++) rgba16f[4 * i + k] = float_to_half(states[i].radiance[k]);
        rgba16f[4 * i + 3] = float_to_half(states[i].alpha);
    }
}

/* The whole Awake() build (Sc/RaytracingMeshDrawer.cs:30-51) on the host, for the CPU baseline:
 * Morton/AABB -> sort -> DistributeKeys -> ConstructTree -> ConstructBVH. */
int orc_build_all(const lbvh_triangle* tris, uint32_t n, uint32_t capacity, const float box_min[3],
                  const float box_max[3], uint32_t* keys, uint32_t* indices, lbvh_aabb* tri_aabb,
                  lbvh_internal_node* internal, lbvh_leaf_node* leaf, lbvh_aabb* bvh, int threads)
{
    if (n < 2 || capacity < n) return -1;
    orc_morton_aabb(tris, n, capacity, box_min, box_max, keys, indices, tri_aabb, threads);
    orc_sort_pairs_mt(keys, indices, capacity, threads);
    orc_distribute_keys_mt(keys, n, threads);
    memset(internal, 0xFF, (size_t)capacity * sizeof *internal);   /* NullLeaf fill :114-115 */
    memset(leaf, 0xFF, (size_t)capacity * sizeof *leaf);
    int rc = orc_build_tree(n, keys, internal, leaf, threads);
    if (rc) return rc;
    return orc_refit_mt(n, internal, leaf, tri_aabb, indices, bvh, threads);
}
