// lbvh_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Replaces ComputeBufferSorter.Sort() (Assets/_Scripts/ComputeBufferSorter.cs:100-126) and its
// kernels LocalRadixSort / PreScan / BlockSum / GlobalScan / GlobalRadixSort
// (Assets/_Shaders/Sorting/*.compute).  Same decomposition of a pass — per-tile digit counts,
// exclusive scan of the (digit, tile) table, scatter — but built for wave64:
//   * a tile is 256 threads x ITEMS keys (4096), not 1024 x 1: digit runs in the scatter are
//     4x longer, so the HBM writes coalesce;
//   * the reference's 8 one-bit split passes with 5 group barriers each (LocalRadixSort.compute:
//     64-91, WavePrefixCountBits over 32 lanes) become ONE ranking step per key: the wave's 64 keys
//     are matched on the whole 8-bit digit with 8 ballots, rank = v_mbcnt of the peer mask, and the
//     per-wave digit counters live in LDS;
//   * the (digit, tile) table is tile-major so every access is a coalesced 1-KB row, and the scan
//     over it is a column scan (three small kernels) instead of the reference's PreScan/BlockSum/
//     GlobalScan over a digit-major table (which needs strided 4-B gathers in the scatter).
// Stability (equal keys keep input order) inside a tile and across tiles makes the output the
// unique stable sort = the reference's result, bit for bit.
#include "lbvh_common.h"

namespace {

constexpr int kThreads = 256;          // 4 waves
constexpr int kWaves = kThreads / LBVH_WAVE;
constexpr int kRadix = 256;            // 8-bit digits, 4 passes (Assets/_Shaders/Constants.cginc:1-2)
constexpr int kItems = 16;             // keys per thread
constexpr int kTile = kThreads * kItems;
constexpr int kChunkTiles = 32;        // tiles per column-scan chunk

// ---- per-tile digit histogram ("upsweep") ---------------------------------------------------
__global__ __launch_bounds__(kThreads) void sort_upsweep_kernel(const uint32_t* __restrict__ keys,
                                                                uint32_t count, uint32_t shift,
                                                                uint32_t* __restrict__ tile_hist)
{
    __shared__ uint32_t s_hist[kWaves][kRadix];
    const uint32_t t = threadIdx.x;
    const uint32_t w = t >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * (uint32_t)kTile;

#pragma unroll
    for (int i = 0; i < kWaves; i++) s_hist[i][t] = 0;
    __syncthreads();

    uint32_t k[kItems];
#pragma unroll
    for (int j = 0; j < kItems; j++) {
        const uint32_t idx = base + (uint32_t)j * kThreads + t;
        k[j] = idx < count ? keys[idx] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kItems; j++) {
        const uint32_t idx = base + (uint32_t)j * kThreads + t;
        const bool valid = idx < count;
        const uint32_t d = valid ? ((k[j] >> shift) & (kRadix - 1)) : 0xFFFFFFFFu;
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        if (__all(d == d0)) {
            // whole wave on one digit (top bytes of Morton codes, pad keys): one plain update
            if (lane_id() == 0 && d0 < (uint32_t)kRadix) s_hist[w][d0] += LBVH_WAVE;
        } else if (valid) {
            atomicAdd(&s_hist[w][d], 1u);
        }
    }
    __syncthreads();
    tile_hist[(size_t)tile * kRadix + t] = s_hist[0][t] + s_hist[1][t] + s_hist[2][t] + s_hist[3][t];
}

// ---- column scan of the tile-major table: H[tile][d] -> global output index of the tile's first
// key with digit d.  Thread d owns column d everywhere.
__global__ __launch_bounds__(kRadix) void sort_scan_reduce_kernel(const uint32_t* __restrict__ tile_hist,
                                                                  uint32_t tiles,
                                                                  uint32_t* __restrict__ chunk_sums)
{
    const uint32_t d = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t t0 = c * kChunkTiles;
    const uint32_t t1 = min(t0 + (uint32_t)kChunkTiles, tiles);
    uint32_t s = 0;
    for (uint32_t t = t0; t < t1; t++) s += tile_hist[(size_t)t * kRadix + d];
    chunk_sums[(size_t)c * kRadix + d] = s;
}

__global__ __launch_bounds__(kRadix) void sort_scan_chunks_kernel(uint32_t* __restrict__ chunk_sums,
                                                                  uint32_t chunks)
{
    __shared__ uint32_t s_wave[kRadix / LBVH_WAVE];
    const uint32_t d = threadIdx.x;
    // exclusive prefix over chunks, per digit
    uint32_t running = 0;
    for (uint32_t c = 0; c < chunks; c++) {
        const uint32_t x = chunk_sums[(size_t)c * kRadix + d];
        chunk_sums[(size_t)c * kRadix + d] = running;
        running += x;
    }
    // exclusive prefix of the digit totals over digits = first output index of each digit
    const uint32_t incl = wave_inclusive_sum(running);
    if ((d & 63) == 63) s_wave[d >> 6] = incl;
    __syncthreads();
    uint32_t wave_prefix = 0;
    for (uint32_t i = 0; i < (d >> 6); i++) wave_prefix += s_wave[i];
    const uint32_t digit_start = incl - running + wave_prefix;
    for (uint32_t c = 0; c < chunks; c++) chunk_sums[(size_t)c * kRadix + d] += digit_start;
}

__global__ __launch_bounds__(kRadix) void sort_scan_apply_kernel(uint32_t* __restrict__ tile_hist,
                                                                 uint32_t tiles,
                                                                 const uint32_t* __restrict__ chunk_sums)
{
    const uint32_t d = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t t0 = c * kChunkTiles;
    const uint32_t t1 = min(t0 + (uint32_t)kChunkTiles, tiles);
    uint32_t running = chunk_sums[(size_t)c * kRadix + d];
    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t x = tile_hist[(size_t)t * kRadix + d];
        tile_hist[(size_t)t * kRadix + d] = running;
        running += x;
    }
}

// ---- rank + scatter ("downsweep") -----------------------------------------------------------
// Lanes of a wave that hold the same 8-bit digit: 8 ballots, one per digit bit.
__device__ __forceinline__ uint64_t match_digit(uint32_t digit)
{
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

__global__ __launch_bounds__(kThreads) void sort_downsweep_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t count, uint32_t shift,
    const uint32_t* __restrict__ tile_base)
{
    __shared__ uint32_t s_keys[kTile];
    __shared__ uint32_t s_vals[kTile];
    __shared__ uint32_t s_wcnt[kWaves][kRadix];  // per-wave digit counts, then per-wave local bases
    __shared__ uint32_t s_gofs[kRadix];          // global base of digit d minus its local start
    __shared__ uint32_t s_wsum[kWaves];

    const uint32_t t = threadIdx.x;
    const uint32_t w = t >> 6;
    const uint32_t lane = lane_id();
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * (uint32_t)kTile;
    const uint32_t nvalid = min((uint32_t)kTile, count - base);

#pragma unroll
    for (int i = 0; i < kWaves; i++) s_wcnt[i][t] = 0;

    // wave-striped load: wave w owns keys [base + w*64*ITEMS, +64*ITEMS), item i = 64 consecutive
    // keys, so (item, lane) order is array order — what stability needs.
    uint32_t key[kItems], val[kItems], rank[kItems];
    const uint32_t wave_base = base + w * (uint32_t)(LBVH_WAVE * kItems);
#pragma unroll
    for (int i = 0; i < kItems; i++) {
        const uint32_t idx = wave_base + (uint32_t)i * LBVH_WAVE + lane;
        const bool valid = idx < count;
        // slots past the end behave as 0xFFFFFFFF keys: they are last in array order and carry the
        // largest digit in every pass, so they rank after every real key and are never written.
        key[i] = valid ? keys_in[idx] : 0xFFFFFFFFu;
        val[i] = valid ? vals_in[idx] : 0xFFFFFFFFu;
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < kItems; i++) {
        const uint32_t d = (key[i] >> shift) & (kRadix - 1);
        const uint64_t peers = match_digit(d);
        const uint32_t r = mbcnt64(peers);              // same-digit lanes below me
        const uint32_t old = s_wcnt[w][d];              // same-digit keys of earlier items (LDS is in
        if (r == 0) s_wcnt[w][d] = old + (uint32_t)__popcll(peers);  // order within a wave)
        rank[i] = old + r;
    }
    __syncthreads();

    {   // thread t = digit t: wave prefix per digit, block scan over digits, global offsets
        const uint32_t c0 = s_wcnt[0][t], c1 = s_wcnt[1][t], c2 = s_wcnt[2][t], c3 = s_wcnt[3][t];
        const uint32_t total = c0 + c1 + c2 + c3;
        const uint32_t incl = wave_inclusive_sum(total);
        if (lane == 63) s_wsum[w] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0;
#pragma unroll
        for (int i = 0; i < kWaves; i++) wave_prefix += (uint32_t)i < w ? s_wsum[i] : 0u;
        const uint32_t dstart = incl - total + wave_prefix;
        s_wcnt[0][t] = dstart;
        s_wcnt[1][t] = dstart + c0;
        s_wcnt[2][t] = dstart + c0 + c1;
        s_wcnt[3][t] = dstart + c0 + c1 + c2;
        s_gofs[t] = tile_base[(size_t)tile * kRadix + t] - dstart;
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < kItems; i++) {
        const uint32_t d = (key[i] >> shift) & (kRadix - 1);
        const uint32_t pos = s_wcnt[w][d] + rank[i];
        s_keys[pos] = key[i];
        s_vals[pos] = val[i];
    }
    __syncthreads();

    // tile is now digit-sorted in LDS: consecutive threads write consecutive addresses inside
    // each digit run.
#pragma unroll
    for (int j = 0; j < kItems; j++) {
        const uint32_t pos = (uint32_t)j * kThreads + t;
        if (pos < nvalid) {
            const uint32_t k = s_keys[pos];
            const uint32_t d = (k >> shift) & (kRadix - 1);
            const uint32_t dst = s_gofs[d] + pos;
            keys_out[dst] = k;
            vals_out[dst] = s_vals[pos];
        }
    }
}

}  // namespace

extern "C" lbvh_status lbvh_sort_pairs(lbvh_context* ctx, uint32_t* d_keys, uint32_t* d_values,
                                       uint32_t count)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (count == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_keys != nullptr && d_values != nullptr);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (count == 1) return LBVH_OK;

    const uint32_t tiles = (uint32_t)(((uint64_t)count + kTile - 1) / kTile);
    const uint32_t chunks = (tiles + kChunkTiles - 1) / kChunkTiles;
    const size_t pair_bytes = (((size_t)count * 4) + 255) & ~(size_t)255;
    const size_t hist_bytes = (size_t)tiles * kRadix * 4;
    const size_t chunk_bytes = (size_t)chunks * kRadix * 4;
    int rc = lbvh_reserve(ctx, &ctx->sort_scratch, &ctx->sort_scratch_bytes,
                          2 * pair_bytes + hist_bytes + chunk_bytes);
    if (rc != LBVH_OK) return rc;
    char* p = (char*)ctx->sort_scratch;
    uint32_t* alt_keys = (uint32_t*)p;
    uint32_t* alt_vals = (uint32_t*)(p + pair_bytes);
    uint32_t* tile_hist = (uint32_t*)(p + 2 * pair_bytes);
    uint32_t* chunk_sums = (uint32_t*)(p + 2 * pair_bytes + hist_bytes);

    uint32_t *ks = d_keys, *vs = d_values, *kd = alt_keys, *vd = alt_vals;
    for (uint32_t shift = 0; shift < 32; shift += 8) {   // ComputeBufferSorter.cs:102
        LBVH_LAUNCH(ctx, sort_upsweep_kernel, dim3(tiles), dim3(kThreads), ks, count,
                           shift, tile_hist);
        LBVH_LAUNCH(ctx, sort_scan_reduce_kernel, dim3(chunks), dim3(kRadix),
                           tile_hist, tiles, chunk_sums);
        LBVH_LAUNCH(ctx, sort_scan_chunks_kernel, dim3(1), dim3(kRadix), chunk_sums,
                           chunks);
        LBVH_LAUNCH(ctx, sort_scan_apply_kernel, dim3(chunks), dim3(kRadix),
                           tile_hist, tiles, chunk_sums);
        LBVH_LAUNCH(ctx, sort_downsweep_kernel, dim3(tiles), dim3(kThreads), ks, vs,
                           kd, vd, count, shift, tile_hist);
        uint32_t* tmp;
        tmp = ks; ks = kd; kd = tmp;
        tmp = vs; vs = vd; vd = tmp;
    }
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;   // 4 passes: the result is back in d_keys / d_values
}
