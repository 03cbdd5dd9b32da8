// Would the LOW passes of an MSD-first radix sort run faster than whole-array passes?  (one MSD pass to
// 256 buckets, then three 8-bit passes per bucket "while it is cache-resident".)  Memory shape only, no ranking: tiles of
// 512 x 16 pairs read in order and written as 256 unaligned 32-key digit runs, 2^26 pairs —
//   G  whole-array pass: the runs of a tile land 2^18 pairs apart (what lbvh_sort's passes do);
//   L  bucket-local pass: the array is 256 buckets of 2^18 pairs (1 MB of keys + 1 MB of values each), a tile's runs stay
//      inside its bucket (1 024 pairs apart); tiles in array order: a bucket's 32 tiles spread over the 8 XCDs;
//   X  the same with every bucket's tiles on ONE XCD (block b runs on XCD b % 8): the bucket's 2 MB + 2 MB stay in one L2;
//   each as ONE pass and as THREE passes back to back ping-ponging between the two buffer pairs (pass k + 1 reads what pass
//   k wrote).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int THREADS = 512, ITEMS = 16, TILE = THREADS * ITEMS;

// MODE 0: global runs; 1: bucket-local, array order; 2: bucket-local, bucket per XCD
template <int MODE>
__global__ __launch_bounds__(THREADS) void pass(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n, uint32_t bucket_tiles)
{
    __shared__ uint32_t s[TILE];
    const uint32_t t = threadIdx.x, w = t >> 6, lane = t & 63;
    uint32_t tile = blockIdx.x;
    if (MODE == 2) {            // blocks b, b + 8, b + 16, ... (one XCD) walk through the tiles of buckets x, x + 8, ...
        const uint32_t x = blockIdx.x & 7u, k = blockIdx.x >> 3;
        tile = ((k / bucket_tiles) * 8u + x) * bucket_tiles + k % bucket_tiles;
    }
    const uint32_t base = tile * TILE, wbase = base + w * 64 * ITEMS;
    uint32_t k[ITEMS], v[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) k[i] = kin[wbase + i * 64 + lane];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) v[i] = vin[wbase + i * 64 + lane];
    uint32_t dst[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) s[w * 64 * ITEMS + i * 64 + lane] = k[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t pos = j * THREADS + t;
        const uint32_t d = pos >> 5, r = pos & 31;                      // 256 runs of 32
        uint32_t o;
        if (MODE == 0) o = (d * (n >> 8) + tile * 32 + r + d * 7 + 13) % n;
        else {
            const uint32_t bucket = tile / bucket_tiles, tb = tile % bucket_tiles, bn = bucket_tiles * TILE;
            o = bucket * bn + (d * (bn >> 8) + tb * 32 + r + d * 7 + 13) % bn;
        }
        dst[j] = o;
        kout[o] = s[pos];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) s[w * 64 * ITEMS + i * 64 + lane] = v[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; j++) vout[dst[j]] = s[j * THREADS + t];
}

int main(int argc, char** argv)
{
    const uint32_t log2n = argc > 1 ? atoi(argv[1]) : 26;
    const uint32_t n = 1u << log2n;
    uint32_t *a, *b, *c, *d;
    CK(hipMalloc(&a, n * 4ull)); CK(hipMalloc(&b, n * 4ull)); CK(hipMalloc(&c, n * 4ull)); CK(hipMalloc(&d, n * 4ull));
    CK(hipMemset(a, 1, n * 4ull)); CK(hipMemset(b, 2, n * 4ull));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned tiles = n / TILE;
    auto run = [&](const char* name, int passes, auto launch) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0));
            for (int p = 0; p < passes; p++) launch(p);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-64s %8.3f ms per pass  %7.1f GB/s\n", name, best / passes, 16.0 * n * passes / best / 1e6);
    };
    for (uint32_t buckets : {256u, 1024u, 64u}) {
        const uint32_t bt = tiles / buckets;
        if (bt == 0) continue;
        printf("-- %u buckets of %u pairs (%u tiles, %.2f MB of keys + values each)\n", buckets, n / buckets, bt, (n / buckets) * 8.0 / 1e6);
        for (int passes : {1, 3}) {
            char nm[96];
            snprintf(nm, 96, "G whole-array runs, %d pass(es)", passes);
            run(nm, passes, [&](int p) { if (p & 1) hipLaunchKernelGGL(pass<0>, dim3(tiles), dim3(THREADS), 0, 0, c, d, a, b, n, bt); else hipLaunchKernelGGL(pass<0>, dim3(tiles), dim3(THREADS), 0, 0, a, b, c, d, n, bt); });
            snprintf(nm, 96, "L bucket-local runs, tiles in array order, %d pass(es)", passes);
            run(nm, passes, [&](int p) { if (p & 1) hipLaunchKernelGGL(pass<1>, dim3(tiles), dim3(THREADS), 0, 0, c, d, a, b, n, bt); else hipLaunchKernelGGL(pass<1>, dim3(tiles), dim3(THREADS), 0, 0, a, b, c, d, n, bt); });
            if (buckets % 8 == 0) {
                snprintf(nm, 96, "X bucket-local runs, a bucket on one XCD, %d pass(es)", passes);
                run(nm, passes, [&](int p) { if (p & 1) hipLaunchKernelGGL(pass<2>, dim3(tiles), dim3(THREADS), 0, 0, c, d, a, b, n, bt); else hipLaunchKernelGGL(pass<2>, dim3(tiles), dim3(THREADS), 0, 0, a, b, c, d, n, bt); });
            }
        }
    }
    return 0;
}
