// What does the onesweep pass's MEMORY SHAPE alone cost?  Tile kernels (512 threads x 16 dwords of keys + 16 of
// values) with no ranking: (A) wave-striped dword loads + dword stores to the same place, (B) through LDS with the
// barriers of the exchange, (C/D) stores following the scatter pattern of uniform 8-bit digits (256 runs per tile,
// aligned 32-dword runs, and unaligned jittered runs).  N = 2^26 pairs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int THREADS = 512, ITEMS = 16, TILE = THREADS * ITEMS;

template <int MODE>
__global__ __launch_bounds__(THREADS) void tilek(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                 uint32_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n,
                                                 uint32_t G = 1)
{
    __shared__ uint32_t s[TILE];
    const uint32_t t = threadIdx.x, w = t >> 6, lane = t & 63;
    // G consecutive tiles per XCD (block b is dispatched to XCD b % 8)
    const uint32_t b_ = blockIdx.x;
    const uint32_t tile = ((b_ >> 3) / G) * (8 * G) + (b_ & 7) * G + (b_ >> 3) % G, base = tile * TILE;
    const uint32_t wbase = base + w * 64 * ITEMS;
    uint32_t k[ITEMS], v[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) k[i] = kin[wbase + i * 64 + lane];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) v[i] = vin[wbase + i * 64 + lane];
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) kout[wbase + i * 64 + lane] = k[i];
#pragma unroll
        for (int i = 0; i < ITEMS; i++) vout[wbase + i * 64 + lane] = v[i];
        return;
    }
    const uint32_t tiles = gridDim.x;
    uint32_t dst[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) s[(w * 64 * ITEMS + i * 64 + lane) ^ 0] = k[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t pos = j * THREADS + t;
        const uint32_t kk = s[pos];
        if (MODE == 1) dst[j] = base + pos;
        else {
            const uint32_t d = pos >> 5, r = pos & 31;                      // 256 runs of 32
            uint32_t o = d * (n >> 8) + tile * 32 + r;
            if (MODE == 3) o = (o + d * 7 + 13) % n;                           // unaligned runs
            dst[j] = o;
        }
        kout[dst[j]] = kk;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) s[w * 64 * ITEMS + i * 64 + lane] = v[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; j++) vout[dst[j]] = s[j * THREADS + t];
    (void)tiles;
}

int main()
{
    const uint32_t n = 1u << 26;
    uint32_t *a, *b, *c, *d;
    CK(hipMalloc(&a, n * 4ull)); CK(hipMalloc(&b, n * 4ull)); CK(hipMalloc(&c, n * 4ull)); CK(hipMalloc(&d, n * 4ull));
    CK(hipMemset(a, 1, n * 4ull)); CK(hipMemset(b, 2, n * 4ull));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        float best = 1e9;
        for (int r = 0; r < 6; r++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-46s %8.3f ms  %7.1f GB/s\n", name, best, 16.0 * n / best / 1e6);
    };
    const unsigned tiles = n / TILE;
    run("A dword striped copy (no LDS)", [&] { hipLaunchKernelGGL(tilek<0>, dim3(tiles), dim3(THREADS), 0, 0, a, b, c, d, n); });
    run("B through LDS, same place", [&] { hipLaunchKernelGGL(tilek<1>, dim3(tiles), dim3(THREADS), 0, 0, a, b, c, d, n); });
    run("C scatter, 256 aligned 128-B runs per tile", [&] { hipLaunchKernelGGL(tilek<2>, dim3(tiles), dim3(THREADS), 0, 0, a, b, c, d, n); });
    run("D scatter, unaligned runs", [&] { hipLaunchKernelGGL(tilek<3>, dim3(tiles), dim3(THREADS), 0, 0, a, b, c, d, n); });
    for (uint32_t G : {1u, 2u, 4u, 16u, 64u}) {
        char nm[64]; snprintf(nm, 64, "D unaligned runs, %u consecutive tiles per XCD", G);
        run(nm, [&] { hipLaunchKernelGGL(tilek<3>, dim3(tiles), dim3(THREADS), 0, 0, a, b, c, d, n, G); });
    }
    return 0;
}
