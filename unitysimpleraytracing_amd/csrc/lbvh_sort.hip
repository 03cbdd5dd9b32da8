// lbvh_sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 ("onesweep" form).
//
// Replaces ComputeBufferSorter.Sort() (Assets/_Scripts/ComputeBufferSorter.cs:100-126) and its
// kernels LocalRadixSort / PreScan / BlockSum / GlobalScan / GlobalRadixSort
// (Assets/_Shaders/Sorting/*.compute).  Same digit width and pass count as the reference (8 bits x 4,
// Assets/_Shaders/Constants.cginc:1-2) and the same per-pass contract — stable partition by digit —
// so the output is the unique stable sort = the reference's result bit for bit.  What changes:
//   * one histogram kernel reads the keys ONCE and produces all four 256-bin digit histograms
//     (the reference recounts per pass inside LocalRadixSort);
//   * each pass is ONE kernel: a tile (512 threads x 8 or 16 keys) ranks its keys, publishes its 256
//     digit counts and obtains the counts of all earlier tiles by decoupled look-back over per-tile
//     status words, instead of the reference's digit-major table + three scan dispatches
//     (Scan.compute:15-96).  16 B/pair/pass + 4 B/pair once, against 20 B/pair/pass;
//   * the reference's 8 one-bit split passes with 5 group barriers each (LocalRadixSort.compute:64-91,
//     WavePrefixCountBits over 32 lanes) become one ranking step per key: the wave's 64 keys are
//     matched on the whole 8-bit digit through an LDS cell per (wave, digit) — every lane ORs its lane
//     bit into the cell's 64-bit peer mask and reads it back; rank = v_mbcnt of the mask + the count
//     the cell keeps of earlier items;
//   * the tile is digit-sorted in LDS and written so consecutive lanes hit consecutive addresses of a
//     digit run (the reference scatters 1024-key tiles: 16-B runs).
// Inter-workgroup protocol (MI355X: 8 XCDs, private non-coherent L2s): a status word carries its own
// flag (2 bits) and value (30 bits), is written with an agent-scope relaxed atomic store
// (global_store_dword sc1, write-through) and polled with agent-scope relaxed atomic loads (sc1, L1
// bypass) — no fences, no ordering between words needed.  Tiles take their index from atomic tickets.
// With ONE ticket queue (tile = ticket) every tile a look-back waits for has been handed out before the
// waiting one: placement-independent by construction.  The 8-queue form below is only selected on the
// layout it was designed for (lbvh_create: all 256 CUs, unmasked stream — there a slot that frees up on an
// XCD is refilled by a workgroup whose home queue is that XCD's, so the lowest tile not yet handed out is
// always the next one somebody takes); on anything else (CPX/DPX/QPX partitions, CU-masked streams) it
// could strand the tiles of queues nobody calls home, so those contexts use one queue.  Every spin is
// bounded (LBVH_SPIN_LIMIT): a protocol failure reports a fault through lbvh_sync, it never hangs the GPU.
// XCD-aware tile order: a digit run written by tile T ends in the 128-B line where tile T+1's run of the
// same digit begins.  If the two tiles run on different XCDs the line is half-written in two L2s and
// reaches memory as two masked partial writes; on one XCD the halves merge in its L2.  So tickets are
// per XCD and hand out kGroup CONSECUTIVE tiles to each XCD in turn (measured on the access pattern
// alone, tools/ubench/tilecopy.hip: 2.9 -> 4.0 TB/s for unaligned 128-B runs).
// Round 5 — the TWO-LEVEL form for latency-bound sizes (2^15 <= count < 2^21: the 1 M-triangle rebuild).  There a pass is not
// bandwidth but the latency of ~250 dependent tiles (ticket, rank, two-level look-back, scatter: 15 us each, 62 us for four).
// Two-level: ONE global onesweep pass that partitions the pairs stably into 256 buckets — balanced ranges of the keys' 12-bit
// prefix (build_bucket_map below; the prefix sits below the caller's key_bits hint: bits 18 .. 29 for Morton codes < 2^30, with
// the 0xFFFFFFFF pads and anything else >= 2^30 in the last bin) — then ONE kernel in which a workgroup sorts its whole bucket by
// key - (the bucket's first prefix) in as many stable 8-bit LSD passes as that difference has bytes (3 for Morton codes), with
// the pairs in registers and the exchanges through LDS, no look-back, no global traffic between the passes.  Two launches and
// one look-back chain instead of five and four.  The result is the same unique stable sort.  A bucket beyond one workgroup's
// registers (16 384 pairs: more than ~12 K pairs with ONE 12-bit prefix) is sorted by its workgroup chunk by chunk through global
// memory: correct for any input, slow — so the form is chosen per call from the LAST sorts' largest buckets (mapped host words
// the first pass kernel leaves behind; a hint one sort stale, like the traversal's dispatch history: either form gives the same
// words).
#include "lbvh_common.h"

#include <algorithm>

namespace {

constexpr int kThreads = 256;          // 4 waves
constexpr int kRadix = 256;
constexpr int kPasses = 4;

constexpr uint32_t kFlagAgg = 1u << 30;    // value = this tile's (this group's) digit count
constexpr uint32_t kFlagIncl = 2u << 30;   // group words: value = digit count of groups 0..this
constexpr uint32_t kValueMask = (1u << 30) - 1u;
constexpr int kLook = 2;                   // group words inspected per look-back step
constexpr int kLbGroup = 8;                // tiles per look-back group

// ---- the two-level form's buckets: BALANCED ranges of a 12-bit key prefix ---------------------------------------------------------
// The bucket digit is not a fixed byte of the key (the top byte of cfg2's Morton codes: largest bucket 11 954 pairs, 3 x the
// mean, and the bucket kernel's run time is its largest bucket's) but a monotone map from the key's 12-bit prefix
// (fine bin = min(key >> fine_shift, 4095), fine_shift = key_bits - 12) to 256 buckets of about count / 256 pairs each:
// bucket(bin) = floor(256 x (pairs in bins before it) / count), from a 4096-bin histogram the histogram kernel takes in the same
// read of the keys.  Every tile of the MSD pass (and every bucket's workgroup) derives the map from those 16 KB by itself:
// cheaper than a launch.  A bucket is then at most count / 256 + its largest bin; it covers the bins [first, last] and is
// sorted by key - (first << fine_shift) in as many 8-bit passes as that difference has bytes (3 for Morton codes: 41 bins x 2^18).
#ifndef LBVH_BUCKET_WAVE_PAIRS
#define LBVH_BUCKET_WAVE_PAIRS 384     // pairs per active wave of the bucket kernel until all sixteen waves are in use
                                       // (cfg2's balanced buckets of 3.9 .. 5.5 K pairs: 256 / 384 / 512 / 768 / 1024 -> 24.6 / 24.0 / 25.5 / 27.3 / 30.2 us)
#endif
#ifndef LBVH_FINE_DEPTH
#define LBVH_FINE_DEPTH 8
#endif
#ifndef LBVH_FINE_BLOCK_KEYS
#define LBVH_FINE_BLOCK_KEYS 8192       // keys per block of the histogram kernel when it takes the fine bins (up to 4096 global atomics per block;
                                        // 4 K x 4, 8 K x 4 / x 8 loads in flight, 16 K x 8: 8.5 / 8.6 / 8.4 / 10.5 us at 1 M keys)
#endif
constexpr int kFineBins = 4096;
constexpr int kFineLog2 = 12;
__device__ __forceinline__ uint32_t fine_bin(uint32_t k, uint32_t fine_shift) { return min(k >> fine_shift, (uint32_t)kFineBins - 1u); }

// ---- all four digit histograms in one read of the keys ------------------------------------------
// HIST_FOUR: ghist[p][256] for the four LSD passes; HIST_FINE: gfine[4096] = the fine bins of the two-level form (the four-pass
// form takes them too where the size allows the other form: the next call's choice is made from what they say)
enum { HIST_FOUR = 1, HIST_FINE = 2 };
template <int MODE>
__global__ __launch_bounds__(kThreads) void sort_histogram_kernel(const uint32_t* __restrict__ keys,
                                                                  uint32_t count, uint32_t* __restrict__ ghist, uint32_t* __restrict__ gfine,
                                                                  uint32_t fine_shift)
{
    __shared__ uint32_t s_hist[(MODE & HIST_FOUR) ? kPasses : 1][kRadix];
    __shared__ uint32_t s_fine[(MODE & HIST_FINE) ? kFineBins : 1];
    const uint32_t t = threadIdx.x;
    if (MODE & HIST_FOUR) {
#pragma unroll
        for (int p = 0; p < kPasses; p++) s_hist[p][t] = 0;
    }
    if (MODE & HIST_FINE) {
#pragma unroll
        for (int i = 0; i < kFineBins / kThreads; i++) s_fine[i * kThreads + t] = 0;
    }
    __syncthreads();
    // grid-stride over 16-B vectors (4 keys per lane per load), 2 loads in flight per thread
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // [head | whole 16-B vectors | tail]: head = keys before the first 16-B boundary (callers may pass any 4-B
    // aligned pointer)
    uint32_t head = (uint32_t)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(keys) & 15u)) & 15u) >> 2);
    if (head > count) head = count;
    const uint32_t nvec = (count - head) >> 2;
    const uint32_t tail0 = head + (nvec << 2);
    const u32x4* vkeys = reinterpret_cast<const u32x4*>(keys + head);
    const uint32_t stride = gridDim.x * kThreads;
    auto add_key = [&](uint32_t k) {
        // a wave whose 64 keys agree on a digit would serialise 64 LDS atomics on one counter (top bytes of
        // clustered Morton codes, 0xFFFFFFFF pads, already-sorted input): one lane adds 64 instead.  Only the two
        // high digits are checked — the low bytes of distinct keys differ — plus whole-key equality.
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        const uint64_t active = __ballot(1);
        const bool first = lane_id() == (uint32_t)__builtin_ctzll(active);
        const uint32_t nactive = (uint32_t)__popcll(active);
        if (__all(k == k0)) {
            if (first) {
                if (MODE & HIST_FOUR) {
#pragma unroll
                    for (int p = 0; p < kPasses; p++) atomicAdd(&s_hist[p][(k0 >> (8 * p)) & 255u], nactive);
                }
                if (MODE & HIST_FINE) atomicAdd(&s_fine[fine_bin(k0, fine_shift)], nactive);
            }
            return;
        }
        if (MODE & HIST_FOUR) {
            atomicAdd(&s_hist[0][k & 255u], 1u);
            atomicAdd(&s_hist[1][(k >> 8) & 255u], 1u);
#pragma unroll
            for (int p = 2; p < kPasses; p++) {
                const uint32_t d = (k >> (8 * p)) & 255u;
                const uint32_t d0 = (k0 >> (8 * p)) & 255u;
                if (__all(d == d0)) {
                    if (first) atomicAdd(&s_hist[p][d0], nactive);
                } else {
                    atomicAdd(&s_hist[p][d], 1u);
                }
            }
        }
        if (MODE & HIST_FINE) {
            const uint32_t d = fine_bin(k, fine_shift), d0 = fine_bin(k0, fine_shift);
            if (__all(d == d0)) {
                if (first) atomicAdd(&s_fine[d0], nactive);
            } else {
                atomicAdd(&s_fine[d], 1u);
            }
        }
    };
    // loads in flight per thread: four 16-byte ones, read as streaming data (the passes read the keys again from memory anyway
    // once they are beyond the L2): 0.092 -> 0.084 ms at 2^26 keys, 10.7 -> 9.7 us at 2^20; eight where the block takes the fine
    // bins alone (fewer, larger blocks: each ends with up to 4096 global atomics)
    constexpr int D = MODE == HIST_FINE ? LBVH_FINE_DEPTH : 4;
    for (uint32_t i0 = blockIdx.x * kThreads + t; i0 < nvec; i0 += stride * D) {
        u32x4 v[D];
        bool ok[D];
#pragma unroll
        for (int k = 0; k < D; k++) {
            ok[k] = i0 + (uint32_t)k * stride < nvec;
            v[k] = u32x4{0u, 0u, 0u, 0u};
            if (ok[k]) v[k] = __builtin_nontemporal_load(&vkeys[i0 + (uint32_t)k * stride]);
        }
#pragma unroll
        for (int k = 0; k < D; k++)
            if (ok[k]) { add_key(v[k].x); add_key(v[k].y); add_key(v[k].z); add_key(v[k].w); }
    }
    if (blockIdx.x == 0 && t < 8u) {                   // at most 3 head + 3 tail keys
        const uint32_t idx = t < 4u ? t : tail0 + (t - 4u);
        if (t < 4u ? t < head : idx < count) {
            const uint32_t k = keys[idx];
            if (MODE & HIST_FOUR) {
#pragma unroll
                for (int p = 0; p < kPasses; p++) atomicAdd(&s_hist[p][(k >> (8 * p)) & 255u], 1u);
            }
            if (MODE & HIST_FINE) atomicAdd(&s_fine[fine_bin(k, fine_shift)], 1u);
        }
    }
    __syncthreads();
    if (MODE & HIST_FOUR) {
#pragma unroll
        for (int p = 0; p < kPasses; p++) {
            const uint32_t c = s_hist[p][t];
            if (c) atomicAdd(&ghist[p * kRadix + t], c);
        }
    }
    if (MODE & HIST_FINE) {
#pragma unroll
        for (int i = 0; i < kFineBins / kThreads; i++) {
            const uint32_t c = s_fine[i * kThreads + t];
            if (c) atomicAdd(&gfine[i * kThreads + t], c);
        }
    }
}

// ---- the bucket map from the fine histogram, by one workgroup of 512 threads (8 bins each) ----------------------------------------
// s_map[bin] = bucket of the bin (monotone); s_bsize / s_bfirst / s_blast[bucket] = its pairs and the first / last NON-EMPTY bin it
// covers (first = 0xFFFFFFFF for a bucket without pairs).  `mul` = floor(2^40 / count): bucket = (pairs before the bin) x 256 / count
// as one v_mul_hi (at most one short of the exact quotient: still monotone, still < 256).
struct bucket_map_lds {
    uint8_t map[kFineBins];
    uint32_t bsize[kRadix], bfirst[kRadix], blast[kRadix];
    uint32_t psize[kRadix];          // the plain map's bucket sizes (16 consecutive bins each)
    uint32_t wsum[8], cost[2][4], plain;
};
template <bool ON> struct bucket_map_slot { bucket_map_lds v; };
template <> struct bucket_map_slot<false> { uint32_t v; };

// 8-bit LSD passes a bucket over the fine bins [first, last] takes: the bytes of its key span (the last bin holds everything from
// 4095 << fine_shift up: 0xFFFFFFFF pads beside Morton codes)
__device__ __forceinline__ uint32_t bucket_passes(uint32_t first_bin, uint32_t last_bin, uint32_t fine_shift, uint32_t& base)
{
    base = first_bin << fine_shift;
    const uint32_t top = last_bin >= (uint32_t)kFineBins - 1u ? 0xFFFFFFFFu : ((last_bin + 1u) << fine_shift) - 1u;
    const uint32_t span = top - base;
    return span == 0u ? 1u : (39u - (uint32_t)__builtin_clz(span)) / 8u;      // 1 .. 4
}

// Two candidate maps, the cheaper one is kept (cost = the largest pairs x passes of any bucket: what the bucket kernel's run time
// follows): the BALANCED one above, and the PLAIN one — bucket = bin >> 4, the top byte of the prefix — which wins where the keys
// are spread evenly already: its buckets are aligned, 16 bins never span more than fine_shift + 4 bits, while a balanced bucket of
// 17 bins of uniform 32-bit keys needs a fourth pass.  A pure function of the histogram: every workgroup reaches the same map.
struct fine_bins8 { uint4 lo, hi; };      // thread t's 8 fine bins, requested early (their latency hides behind the tile's key loads)
__device__ __forceinline__ fine_bins8 load_fine_bins(const uint32_t* __restrict__ gfine)
{
    fine_bins8 r;
    r.lo = reinterpret_cast<const uint4*>(gfine)[2u * threadIdx.x];
    r.hi = reinterpret_cast<const uint4*>(gfine)[2u * threadIdx.x + 1u];
    return r;
}
__device__ __forceinline__ void build_bucket_map(const fine_bins8& fb, uint32_t mul, uint32_t fine_shift, bucket_map_lds& L)
{
    const uint32_t t = threadIdx.x, w = t >> 6, lane = lane_id();     // blockDim.x == 512
    if (t < (uint32_t)kRadix) { L.bsize[t] = 0u; L.bfirst[t] = 0xFFFFFFFFu; L.blast[t] = 0u; L.psize[t] = 0u; }
    const uint32_t f[8] = {fb.lo.x, fb.lo.y, fb.lo.z, fb.lo.w, fb.hi.x, fb.hi.y, fb.hi.z, fb.hi.w};
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) mine += f[j];
    const uint32_t incl = wave_inclusive_sum(mine);
    if (lane == 63) L.wsum[w] = incl;
    __syncthreads();                                  // also: the tables' initial values are in place
    uint32_t before = incl - mine;
#pragma unroll
    for (int i = 0; i < 8; i++) before += (uint32_t)i < w ? L.wsum[i] : 0u;
    if (mine != 0) atomicAdd(&L.psize[t >> 1], mine);            // my 8 bins are half of plain bucket t / 2
    // my 8 bins: runs of equal bucket go to the tables with one atomic each
    uint32_t bytes[2] = {0u, 0u};
    uint32_t cur = 0xFFFFFFFFu, acc = 0, lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t bucket = min(__umulhi(before, mul), (uint32_t)kRadix - 1u);
        bytes[j >> 2] |= bucket << (8 * (j & 3));
        if (bucket != cur) {
            if (acc != 0) { atomicAdd(&L.bsize[cur], acc); atomicMin(&L.bfirst[cur], lo); atomicMax(&L.blast[cur], hi); }
            cur = bucket; acc = 0; lo = 0xFFFFFFFFu; hi = 0;
        }
        if (f[j] != 0) {
            const uint32_t bin = 8u * t + (uint32_t)j;
            acc += f[j];
            lo = min(lo, bin);
            hi = bin;
        }
        before += f[j];
    }
    if (acc != 0) { atomicAdd(&L.bsize[cur], acc); atomicMin(&L.bfirst[cur], lo); atomicMax(&L.blast[cur], hi); }
    __syncthreads();
    if (t < (uint32_t)kRadix) {                       // thread t = bucket t: the two maps' costs
        uint32_t base;
        uint32_t cb = L.bsize[t] != 0u ? L.bsize[t] * bucket_passes(L.bfirst[t], L.blast[t], fine_shift, base) : 0u;
        uint32_t cp = L.psize[t] * bucket_passes(16u * t, 16u * t + 15u, fine_shift, base);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            cb = max(cb, (uint32_t)__shfl_xor((int)cb, d));
            cp = max(cp, (uint32_t)__shfl_xor((int)cp, d));
        }
        if (lane == 0) { L.cost[0][w] = cb; L.cost[1][w] = cp; }
    }
    __syncthreads();
    const uint32_t cost_b = max(max(L.cost[0][0], L.cost[0][1]), max(L.cost[0][2], L.cost[0][3]));
    const uint32_t cost_p = max(max(L.cost[1][0], L.cost[1][1]), max(L.cost[1][2], L.cost[1][3]));
    const bool plain = cost_p <= cost_b;              // uniform over the workgroup (and over the grid)
    if (plain) {
        bytes[0] = bytes[1] = (t >> 1) * 0x01010101u;
        if (t < (uint32_t)kRadix) { L.bsize[t] = L.psize[t]; L.bfirst[t] = L.psize[t] != 0u ? 16u * t : 0xFFFFFFFFu; L.blast[t] = 16u * t + 15u; }
    }
    reinterpret_cast<uint32_t*>(L.map)[2u * t] = bytes[0];
    reinterpret_cast<uint32_t*>(L.map)[2u * t + 1u] = bytes[1];
    __syncthreads();
}

// the largest bucket of this input -> 4 mapped host words (one per wave of buckets; the host takes their maximum), each tagged
// with the fine shift it belongs to: the NEXT sort's choice of form (lbvh_launch_sort)
__device__ __forceinline__ void publish_bucket_stat(const bucket_map_lds& L, uint32_t* bucket_stat, uint32_t fine_shift)
{
    const uint32_t t = threadIdx.x, w = t >> 6, lane = lane_id();
    if (t < (uint32_t)kRadix) {
        uint32_t m = L.bsize[t];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
        if (lane == 0) __hip_atomic_store(bucket_stat + w, (fine_shift << 24) | min(m, 0xFFFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Ticket k of queue x -> tile.  queues == 8: queue x holds the tiles whose (tile / group) % 8 == x in increasing
// order (`group` consecutive tiles per XCD in turn); queues == 1: tile = ticket.
__host__ __device__ __forceinline__ uint32_t ticket_tile(uint32_t k, uint32_t x, uint32_t group, uint32_t queues)
{
    return (k / group) * (queues * group) + x * group + (k % group);
}

// ---- one pass: rank + look-back + scatter ----------------------------------------------------------
// MSD: the two-level form's one global pass — digit = the key's bucket (build_bucket_map) instead of a byte of the key;
// `shift` is then the fine shift
template <int THREADS, int ITEMS, bool STREAM, bool MSD = false>
__global__ __launch_bounds__(THREADS) void sort_onesweep_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t count, uint32_t shift,
    const uint32_t* __restrict__ ghist,   // [256] digit totals of this pass (MSD: unused — the bucket sizes come out of the map)
    const uint32_t* __restrict__ gfine,   // (nullable; 512-thread x 8-item form only) [4096] fine bins: MSD derives its buckets from them;
                                          // otherwise tile 0 derives them after its own work, for ...
    uint32_t* bucket_stat,                // ... these 4 mapped host words, each (fine shift << 24 | largest bucket): the NEXT sort's choice of form
    uint32_t fine_shift, uint32_t fine_mul,   // fine bin = min(key >> fine_shift, 4095); fine_mul = floor(2^40 / count)
    uint32_t* __restrict__ gtable,        // MSD: [3][256] bucket sizes, first and last fine bin — tile 0 writes them for the bucket kernel
    uint32_t* status,                     // [tiles][256] tile words of this pass (zeroed per sort)
    uint32_t* gstatus,                    // [ceil(tiles / kLbGroup)][256] group words of this pass (zeroed per sort)
    uint32_t* tickets,                    // [8] per-XCD tile tickets of this pass (zeroed per sort)
    uint32_t tiles, uint32_t group,       // group = consecutive tiles handed to one XCD
    uint32_t queues,                      // 8 = per-XCD ticket queues, 1 = tiles in ticket order
    uint32_t* fault)                      // mapped host word: a bounded spin that gave up says so here
{
    constexpr int TILE = THREADS * ITEMS;
    // STREAM (sorts from 8 M pairs: beyond what the L2s hold): every key and value is read exactly once per pass, and
    // loaded as streaming data (sc1 nt: not kept in L2) they leave the L2 to the scatter's partial lines, which wait
    // there for the tile that completes them — 1.24 -> 1.19 ms per 2^26 pairs, 0.367 -> 0.346 at 2^24; at 2^22 (32 MB of
    // pairs) the default policy is the faster one (0.129 against 0.134 ms).
    constexpr int kLoadPolicy = STREAM ? 6 : 0;
    constexpr int WAVES = THREADS / LBVH_WAVE;
    constexpr int DWAVES = kRadix / LBVH_WAVE;   // waves that own the 256 digits
    // tile exchange buffer: keys first, then values; during ranking it holds the per-wave rank cells
    constexpr int XCHG_WORDS = TILE > WAVES * kRadix * 4 ? TILE : WAVES * kRadix * 4;
    __shared__ __attribute__((aligned(16))) uint32_t s_xchg[XCHG_WORDS];
    __shared__ uint16_t s_wcnt[WAVES][kRadix];   // per-wave local bases (<= TILE <= 8192)
    __shared__ uint32_t s_gofs[kRadix];          // global base of digit d minus its local start
    __shared__ uint32_t s_wsum[DWAVES + 1];
    __shared__ uint32_t s_tile;
    constexpr bool kMapCapable = THREADS == 512 && ITEMS == 8 && !STREAM;      // the form used below 2^21 pairs
    static_assert(kMapCapable || !MSD, "the MSD pass is the 512 x 8 form");
    // the MSD pass looks every key's bucket up in the map: its own 8.2 KB.  The LSD passes build a map only in tile 0 of the first
    // pass, AFTER the tile's work (the next sort's hint): there it lives in the exchange buffer, which is free by then — the four
    // passes' workgroups no longer carry 8.2 KB they never use
    __shared__ bucket_map_slot<MSD> s_bslot;                                   // (4 bytes in the LSD passes)
    static_assert(!kMapCapable || sizeof(bucket_map_lds) <= sizeof(uint32_t) * XCHG_WORDS, "the hint's map fits the exchange buffer");

    const uint32_t t = threadIdx.x;
    const uint32_t w = t >> 6;
    const uint32_t lane = lane_id();
    fine_bins8 fine8 = {};
    if constexpr (MSD) fine8 = load_fine_bins(gfine);      // 16 KB per tile from L2, in flight while the tile takes its ticket
    if (t == 0) {
        // when the home queue is drained take from the others.  grid == tiles and every workgroup takes exactly
        // one, so one is found.
        const uint32_t home = xcc_id() & (queues - 1u);
        uint32_t tile = 0;
        for (uint32_t a = 0; a < queues; a++) {
            const uint32_t x = (home + a) & (queues - 1u);
            const uint32_t k = __hip_atomic_fetch_add(tickets + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tile = ticket_tile(k, x, group, queues);
            if (tile < tiles) break;
        }
        s_tile = tile;
    }
    // rank cells {peer mask lo, hi, count, -} per (wave, digit), 16 B each, all zero
    for (int i = t; i < WAVES * kRadix; i += THREADS) reinterpret_cast<uint4*>(s_xchg)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * (uint32_t)TILE;
    const uint32_t nvalid = min((uint32_t)TILE, count - base);
    auto digit_of = [shift](uint32_t k) -> uint32_t {
        if constexpr (MSD) return (uint32_t)s_bslot.v.map[fine_bin(k, shift)];
        else return (k >> shift) & (uint32_t)(kRadix - 1);
    };
    // wave-striped load: wave w owns keys [base + w*64*ITEMS, +64*ITEMS), item i = 64 consecutive
    // keys, so (item, lane) order is array order — what stability needs.
    // Register budget: 64 VGPRs = 8 waves per SIMD = 4 tiles per CU (the look-back and the loads are latency,
    // and residency is what hides it).  Hence ranks / local positions are kept two per register (13 bits each)
    // and the digits of the sorted keys four per register instead of 16 destination indices.
    uint32_t key[ITEMS], rank2[ITEMS / 2];
    const uint32_t wave_base = base + w * (uint32_t)(LBVH_WAVE * ITEMS);
    // Buffer loads: out-of-range lanes of the last tile read 0 (range-checked against count) and are then set to
    // 0xFFFFFFFF — slots past the end behave as the largest key: last in array order, largest digit in every
    // pass, so they rank after every real key and are never written.  No branch, so the ranking of item 0
    // starts when its load lands instead of waiting for all sixteen.
    const __amdgpu_buffer_rsrc_t keys_in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(keys_in), 0, (int)(count * 4u), 0x00020000);
    auto load_keys = [&]() {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const uint32_t idx = wave_base + (uint32_t)i * LBVH_WAVE + lane;
            const uint32_t k = __builtin_amdgcn_raw_buffer_load_b32(keys_in_rsrc, idx * 4u, 0, kLoadPolicy);
            key[i] = idx < count ? k : 0xFFFFFFFFu;
        }
    };
    if constexpr (MSD) {
        // every tile derives the bucket map from the 16 KB of fine bins by itself (cheaper than a launch), with its keys already
        // requested; tile 0 leaves the bucket table to the bucket kernel and the largest bucket to the next call's choice of form
        load_keys();
        build_bucket_map(fine8, fine_mul, fine_shift, s_bslot.v);
        if (tile == 0) {
            if (t < (uint32_t)kRadix) {
                gtable[t] = s_bslot.v.bsize[t];
                gtable[kRadix + t] = s_bslot.v.bfirst[t];
                gtable[2 * kRadix + t] = s_bslot.v.blast[t];
            }
            publish_bucket_stat(s_bslot.v, bucket_stat, fine_shift);
        }
    }

    // exclusive scan of the pass's digit totals = first output index of each digit (every tile
    // recomputes it from 1 KB of L2-resident counters: cheaper than a launch)
    uint32_t digit_start = 0;
    {
        uint32_t total = 0u;
        if constexpr (MSD) total = t < (uint32_t)kRadix ? s_bslot.v.bsize[t] : 0u;
        else total = t < (uint32_t)kRadix ? ghist[t] : 0u;
        const uint32_t incl = wave_inclusive_sum(total);
        if (lane == 63 && w < (uint32_t)DWAVES) s_wsum[w] = incl;
        __syncthreads();
        uint32_t wave_prefix = 0;
#pragma unroll
        for (int i = 0; i < DWAVES; i++) wave_prefix += (uint32_t)i < w ? s_wsum[i] : 0u;
        digit_start = incl - total + wave_prefix;
    }

    if constexpr (!MSD) load_keys();

    // Ranking.  The wave's 64 keys of one item are matched on the whole 8-bit digit THROUGH LDS: every lane ORs its
    // lane bit into the 64-bit peer mask of cell (wave, digit) and reads the cell back — LDS executes a wave's
    // instructions in order, so the read sees all 64 ORs.  rank = count of the same digit in earlier items (kept in
    // the cell) + same-digit lanes below me (v_mbcnt of the mask); the lowest peer lane clears the mask and adds
    // the group size to the count.  12 VALU instructions per key instead of 82 for the 8-ballot form: the pass
    // was as much VALU-bound (196 us of wave64 issue per 2^26 pairs) as HBM-bound.
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4* cells = reinterpret_cast<u32x4*>(s_xchg) + w * kRadix;
        const unsigned long long lane_bit = 1ull << lane;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const uint32_t d = digit_of(key[i]);
            u32x4* cell = cells + d;
            __hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(cell), lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const u32x4 c = *reinterpret_cast<volatile u32x4*>(cell);
            const uint64_t peers = ((uint64_t)c.y << 32) | c.x;
            const uint32_t r = mbcnt64(peers);              // same-digit lanes below me
            const uint32_t old = c.z;                       // same-digit keys of earlier items
            if (r == 0) *cell = u32x4{0u, 0u, old + (uint32_t)__popcll(peers), 0u};
            if (i & 1) rank2[i / 2] |= (old + r) << 16; else rank2[i / 2] = old + r;
            // pin the packed value here: otherwise the compiler sinks the adds to the first use and carries
            // `old` and `r` of all 16 items (32 registers) across the phase instead of 8
            asm volatile("" : "+v"(rank2[i / 2]));
        }
    }
    __syncthreads();

    uint32_t ltotal = 0;
    uint32_t gv[kLook] = {};                           // look-back state carried across the LDS exchange
    uint32_t lb_group = 0, lb_partial = 0, lb_total = 0;
    bool lb_leader = false;
    if (t < (uint32_t)kRadix) {   // thread t = digit t
        uint32_t total = 0;
#pragma unroll
        for (int i = 0; i < WAVES; i++) total += s_xchg[(i * kRadix + t) * 4 + 2];
        ltotal = total;                                // including padding slots (they sit last)
        // the padding slots of a partial last tile (key 0xFFFFFFFF) all landed on the LARGEST digit in use — 255, or in the MSD pass
        // the bucket of the last fine bin (the map is monotone: no key has a larger one, so they still sit last in the tile) —:
        // they are not keys
        uint32_t pad_digit = (uint32_t)kRadix - 1u;
        if constexpr (MSD) pad_digit = s_bslot.v.map[kFineBins - 1];
        if (t == pad_digit) total -= (uint32_t)TILE - nvalid;

        // Two-level look-back.  A tile publishes its digit counts (tile words) and needs the counts of all
        // earlier tiles.  With single-level decoupled look-back the walk length is (tiles finishing per
        // round trip to the coherence point) ~ 30 words on this chip (41 tiles/us x 0.7 us), 14 dependent
        // round trips per tile (measured) — the pass ran at 0.36 ms against 0.22 ms without any look-back.
        // Here tiles form groups of kLbGroup: tile i of group g sums (a) the i earlier tile words of its own
        // group, all requested at once, and (b) the groups before g by a decoupled look-back over GROUP
        // words (aggregate, then inclusive prefix), which only the last tile of each group publishes:
        // ~2 dependent round trips per tile.
        constexpr uint32_t G = (uint32_t)kLbGroup;
        const uint32_t g = tile / G, gi = tile % G;
        __hip_atomic_store(status + (size_t)tile * kRadix + t, kFlagAgg | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t in_group = 0;
        {
            const uint32_t* row0 = status + (size_t)(g * G) * kRadix + t;
            uint32_t v[kLbGroup - 1];
#pragma unroll
            for (int j = 0; j < kLbGroup - 1; j++)
                if ((uint32_t)j < gi) v[j] = __hip_atomic_load(row0 + (size_t)j * kRadix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < kLbGroup - 1; j++) {
                if ((uint32_t)j >= gi) continue;
                for (uint32_t spins = 0; (v[j] & ~kValueMask) == 0; spins++) {
                    if (spins > LBVH_SPIN_LIMIT) {
                        __hip_atomic_store(fault, LBVH_FAULT_SORT_LOOKBACK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                    v[j] = __hip_atomic_load(row0 + (size_t)j * kRadix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                in_group += v[j] & kValueMask;
            }
        }
        const bool leader = gi == G - 1u;               // a partial last group has no leader: nothing follows it
        uint32_t* gmine = gstatus + (size_t)g * kRadix + t;
        if (leader)
            __hip_atomic_store(gmine, (g == 0 ? kFlagIncl : kFlagAgg) | ((in_group + total) & kValueMask), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // the group-level words are requested now and consumed after the tile's LDS exchange below: the
        // round trips to the coherence point overlap with work that does not need the global offsets
#pragma unroll
        for (int j = 0; j < kLook; j++) {
            const uint32_t q = g - 1u - (uint32_t)j;
            gv[j] = (uint32_t)j < g ? __hip_atomic_load(gstatus + (size_t)q * kRadix + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : kFlagIncl;         // before group 0: inclusive prefix 0
        }
        lb_group = g;
        lb_leader = leader;
        lb_partial = in_group;
        lb_total = total;
    }
    uint32_t gbase = 0;                                // first output index of digit t minus its local start
    {   // local layout: digits in order, waves in order inside a digit
        const uint32_t incl = wave_inclusive_sum(ltotal);
        __syncthreads();                               // s_wsum reuse
        if (lane == 63 && w < (uint32_t)DWAVES) s_wsum[w] = incl;
        __syncthreads();
        if (t < (uint32_t)kRadix) {
            uint32_t wave_prefix = 0;
#pragma unroll
            for (int i = 0; i < DWAVES; i++) wave_prefix += (uint32_t)i < w ? s_wsum[i] : 0u;
            const uint32_t dstart = incl - ltotal + wave_prefix;
            uint32_t run = dstart;
#pragma unroll
            for (int i = 0; i < WAVES; i++) {
                const uint32_t c = s_xchg[(i * kRadix + t) * 4 + 2];
                s_wcnt[i][t] = (uint16_t)run;
                run += c;
            }
            gbase = digit_start - dstart;
        }
    }
    __syncthreads();

    // Keys and values go through the SAME LDS buffer one after the other: half the LDS per tile.
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const uint32_t d = digit_of(key[i]);
        const uint32_t r = (i & 1) ? rank2[i / 2] >> 16 : rank2[i / 2] & 0xFFFFu;
        const uint32_t lpos = (uint32_t)s_wcnt[w][d] + r;            // local position in the digit-sorted tile
        if (i & 1) rank2[i / 2] = (rank2[i / 2] & 0xFFFFu) | (lpos << 16); else rank2[i / 2] = (rank2[i / 2] & 0xFFFF0000u) | lpos;
        s_xchg[lpos] = key[i];
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four LDS round trips in flight, not sixteen (registers)
    }
    // values are requested now (the key registers are free) and land while the keys are written out
    uint32_t val[ITEMS];
    {
        const __amdgpu_buffer_rsrc_t vals_in_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(vals_in), 0, (int)(count * 4u), 0x00020000);
#pragma unroll
        for (int i = 0; i < ITEMS; i++)
            val[i] = __builtin_amdgcn_raw_buffer_load_b32(vals_in_rsrc, (wave_base + (uint32_t)i * LBVH_WAVE + lane) * 4u, 0, kLoadPolicy);
    }
    if (t < (uint32_t)kRadix) {   // finish the look-back: decoupled walk over the group words, nearest first
        uint32_t before = 0;
        uint32_t p = lb_group;                       // next word to consume belongs to group p - 1
        bool done = p == 0;
        for (uint32_t spins = 0; !done; spins++) {
            if (spins > LBVH_SPIN_LIMIT) {
                __hip_atomic_store(fault, LBVH_FAULT_SORT_LOOKBACK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
#pragma unroll
            for (int j = 0; j < kLook; j++) {
                if (done) continue;
                const uint32_t f = gv[j] & ~kValueMask;
                if (f == 0) break;                   // not published yet: re-read from here
                before += gv[j] & kValueMask;
                p--;
                if (f == kFlagIncl) done = true;
            }
            if (!done) {
#pragma unroll
                for (int j = 0; j < kLook; j++) {
                    const uint32_t q = p - 1u - (uint32_t)j;
                    gv[j] = (uint32_t)j < p ? __hip_atomic_load(gstatus + (size_t)q * kRadix + t, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT)
                                            : kFlagIncl;
                }
            }
        }
        if (lb_leader && lb_group > 0)
            __hip_atomic_store(gstatus + (size_t)lb_group * kRadix + t, kFlagIncl | ((before + lb_partial + lb_total) & kValueMask),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_gofs[t] = gbase + before + lb_partial;
    }
    __syncthreads();

    // the tile's keys are digit-sorted in LDS: consecutive threads write consecutive addresses inside
    // each digit run.  Buffer stores: a 32-bit byte offset per store instead of a 64-bit address
    // (count < 2^30, so offsets fit), which is what keeps 16 stores in flight inside the register budget.
    const __amdgpu_buffer_rsrc_t keys_rsrc = __builtin_amdgcn_make_buffer_rsrc(keys_out, 0, (int)(count * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t vals_rsrc = __builtin_amdgcn_make_buffer_rsrc(vals_out, 0, (int)(count * 4u), 0x00020000);
    uint32_t dig4[ITEMS / 4];
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t pos = (uint32_t)j * THREADS + t;
        const uint32_t k = s_xchg[pos];
        const uint32_t d = digit_of(k);
        if (j & 3) dig4[j / 4] |= d << (8 * (j & 3)); else dig4[j / 4] = d;
        const uint32_t dst = s_gofs[d] + pos;
        if (pos < nvalid) __builtin_amdgcn_raw_buffer_store_b32(k, keys_rsrc, dst * 4u, 0, 0);
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) s_xchg[(i & 1) ? rank2[i / 2] >> 16 : rank2[i / 2] & 0xFFFFu] = val[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint32_t pos = (uint32_t)j * THREADS + t;
        const uint32_t dst = s_gofs[(dig4[j / 4] >> (8 * (j & 3))) & 255u] + pos;
        if (pos < nvalid) __builtin_amdgcn_raw_buffer_store_b32(s_xchg[pos], vals_rsrc, dst * 4u, 0, 0);
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (kMapCapable && !MSD) {
        // the four-pass form, first pass: what the two-level form's largest bucket WOULD be for this input (after this tile's own
        // work: nobody waits for it)
        if (gfine != nullptr && tile == 0) {
            __syncthreads();                 // (every thread's last read of the exchange buffer is behind it)
            bucket_map_lds& hint_map = *reinterpret_cast<bucket_map_lds*>(s_xchg);
            build_bucket_map(load_fine_bins(gfine), fine_mul, fine_shift, hint_map);
            publish_bucket_stat(hint_map, bucket_stat, fine_shift);
        }
    }
}

#ifndef LBVH_RANK_GROUP
#define LBVH_RANK_GROUP 1          // items of a wave ranked at once by the bucket kernel (bucket_rank; 1 / 2 / 4 measured equal)
#endif
#ifdef LBVH_BUCKET_TIMING
// measurement variant (tools/build_variant.sh timing -DLBVH_BUCKET_TIMING): s_memtime at the phase borders of every bucket
__device__ unsigned long long g_bucket_timing[kRadix][40];
#define LBVH_BT(slot) do { if (threadIdx.x == 0) g_bucket_timing[blockIdx.x][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LBVH_BT(slot) do { } while (0)
#endif

// ---- the two-level form's second kernel: one workgroup sorts one bucket ---------------------------------------------------------
// A tile of up to THREADS x ITEMS pairs held in registers, wave-striped over the first `aw` waves of the workgroup: wave w < aw
// owns slots [w 64 it, (w + 1) 64 it), item i = 64 consecutive slots, so (wave, item, lane) order is array order; the other waves
// hold nothing (a small bucket spread over all sixteen waves pays sixteen waves' worth of rank cells to clear and to scan: the
// kernel's floor was 17 us for 512-pair buckets that way).  bucket_rank ranks the tile by the byte at `pshift` exactly as a pass
// kernel does (per-wave peer-mask cells in LDS) and returns each item's position in the digit-sorted tile (two per register);
// thread t < 256 also gets digit t's count.  Slots past the tile's last pair carry 0xFFFFFFFF: largest digit, last in array
// order, so they rank behind every real pair in every pass.
template <int THREADS, int ITEMS>
__device__ __forceinline__ void bucket_rank(const uint32_t (&key)[ITEMS], uint32_t aw, uint32_t my_it, uint32_t pshift, uint32_t* s_xchg,
                                            uint32_t* s_cnt, uint16_t (*s_wcnt)[kRadix], uint32_t* s_dstart, uint32_t* s_wsum,
                                            uint32_t (&lpos2)[ITEMS / 2], uint32_t& digit_total)
{
    constexpr int DWAVES = kRadix / LBVH_WAVE, WAVES = THREADS / LBVH_WAVE;
    constexpr int G = ITEMS < LBVH_RANK_GROUP ? ITEMS : LBVH_RANK_GROUP;      // items of a wave ranked at once
    const uint32_t t = threadIdx.x, w = t >> 6, lane = lane_id();
    // Ranking: the pass kernels' form with the cell split in two arrays — OR my lane bit into the 64-bit peer mask of (wave,
    // digit), read it back (LDS executes a wave's instructions in order: the read sees all 64 ORs); the lowest peer clears the
    // mask and adds the group's size to the count of (wave, digit) with a RETURNING add, and hands the old count to its peers by
    // a lane permute.  G items may go through that chain together (G banks of masks).  Measured (profiles/r5/b_*): the kernel is
    // bound by the LDS instruction rate of its ONE workgroup per CU — ~940 cycles per item and pass for sixteen waves' ORs, reads,
    // adds and permutes — not by the chain's latency: G = 1 / 2 / 4 run the 12 K-pair bucket in 37.0 / 38.2 / 39.9 us, the
    // 16-byte-cell form of the pass kernels in 38.5.
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(s_xchg);           // [G][WAVES][256]
    for (uint32_t i = t; i < aw * (uint32_t)kRadix; i += THREADS) {
#pragma unroll
        for (int j = 0; j < G; j++) masks[(uint32_t)j * (WAVES * kRadix) + i] = 0ull;
        s_cnt[i] = 0u;
    }
    __syncthreads();
    LBVH_BT(32);
    {
        unsigned long long* wmask = masks + w * kRadix;
        uint32_t* wcount = s_cnt + w * kRadix;
        const unsigned long long lane_bit = 1ull << lane;
#pragma unroll
        for (int g = 0; g < ITEMS; g += G) {
            if ((uint32_t)g >= my_it) continue;                 // uniform over the wave
            uint32_t d[G];
            uint64_t peers[G];
            uint32_t old[G];
#pragma unroll
            for (int j = 0; j < G; j++) {
                if ((uint32_t)(g + j) >= my_it) continue;
                d[j] = (key[g + j] >> pshift) & (kRadix - 1);
                __hip_atomic_fetch_or(wmask + j * (WAVES * kRadix) + d[j], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int j = 0; j < G; j++) {
                if ((uint32_t)(g + j) >= my_it) continue;
                peers[j] = *reinterpret_cast<volatile unsigned long long*>(wmask + j * (WAVES * kRadix) + d[j]);
            }
#pragma unroll
            for (int j = 0; j < G; j++) {
                if ((uint32_t)(g + j) >= my_it) continue;
                old[j] = 0;
                if (mbcnt64(peers[j]) == 0) {                   // the lowest lane of this (item, digit)
                    wmask[j * (WAVES * kRadix) + d[j]] = 0ull;
                    old[j] = __hip_atomic_fetch_add(wcount + d[j], (uint32_t)__popcll(peers[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
#pragma unroll
            for (int j = 0; j < G; j++) {
                if ((uint32_t)(g + j) >= my_it) continue;
                const int i = g + j;
                const uint32_t first = (uint32_t)__shfl((int)old[j], __builtin_ctzll(peers[j]));      // same-digit pairs of earlier items
                const uint32_t r = first + mbcnt64(peers[j]);
                if (i & 1) lpos2[i / 2] |= r << 16; else lpos2[i / 2] = r;
            }
        }
    }
    __syncthreads();
    LBVH_BT(33);
    // digit t: its count, and where each wave's run of it starts inside the digit (ONE serial walk over the active waves)
    uint32_t ltotal = 0;
    if (t < (uint32_t)kRadix) {
        for (uint32_t i = 0; i < aw; i++) {
            const uint32_t c = s_cnt[i * kRadix + t];
            s_wcnt[i][t] = (uint16_t)ltotal;
            ltotal += c;
        }
    }
    const uint32_t incl = wave_inclusive_sum(ltotal);
    if (lane == 63 && w < (uint32_t)DWAVES) s_wsum[w] = incl;
    __syncthreads();
    digit_total = ltotal;
    if (t < (uint32_t)kRadix) {
        uint32_t wave_prefix = 0;
#pragma unroll
        for (int i = 0; i < DWAVES; i++) wave_prefix += (uint32_t)i < w ? s_wsum[i] : 0u;
        s_dstart[t] = incl - ltotal + wave_prefix;
    }
    __syncthreads();
    LBVH_BT(34);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        if ((uint32_t)i >= my_it) continue;
        const uint32_t d = (key[i] >> pshift) & (kRadix - 1);
        const uint32_t r = (i & 1) ? lpos2[i / 2] >> 16 : lpos2[i / 2] & 0xFFFFu;
        const uint32_t lpos = s_dstart[d] + (uint32_t)s_wcnt[w][d] + r;
        if (i & 1) lpos2[i / 2] = (lpos2[i / 2] & 0xFFFFu) | (lpos << 16); else lpos2[i / 2] = (lpos2[i / 2] & 0xFFFF0000u) | lpos;
    }
    // (the callers' next barrier separates these reads of the cells / s_wcnt / s_dstart from whatever overwrites them)
}


template <int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS) void sort_bucket_kernel(uint32_t* __restrict__ keys_in, uint32_t* __restrict__ vals_in,
                                                              uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                              uint32_t fine_shift,
                                                              const uint32_t* __restrict__ gtable)     // [3][256]: bucket sizes, first, last fine bin
{
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVES = THREADS / LBVH_WAVE, DWAVES = kRadix / LBVH_WAVE;
    // (key, value) pairs change places through LDS as 8-byte words: one write, one barrier, one read per pass; the four banks of
    // peer masks (bucket_rank) live in the same buffer between the exchanges
    constexpr int XCHG_WORDS = 2 * TILE > WAVES * kRadix * 2 * LBVH_RANK_GROUP ? 2 * TILE : WAVES * kRadix * 2 * LBVH_RANK_GROUP;
    static_assert(TILE <= 65536, "local positions are kept in 16 bits");
    __shared__ __attribute__((aligned(16))) uint32_t s_xchg[XCHG_WORDS];
    __shared__ uint32_t s_cnt[WAVES][kRadix];    // pairs of (wave, digit) ranked so far in this pass
    __shared__ uint16_t s_wcnt[WAVES][kRadix];
    __shared__ uint32_t s_dstart[kRadix];
    __shared__ uint32_t s_wsum[DWAVES + 1];
    __shared__ uint32_t s_base[kRadix];          // big buckets: next output index of each digit, relative to the bucket
    __shared__ uint32_t s_start, s_size, s_same;
    uint2* s_pair = reinterpret_cast<uint2*>(s_xchg);
    const uint32_t t = threadIdx.x, w = t >> 6, lane = lane_id();
    const uint32_t b = blockIdx.x;
    LBVH_BT(0);
    {   // where bucket b starts: exclusive scan of the 256 bucket sizes (1 KB of L2-resident counters, cheaper than a launch)
        const uint32_t total = t < (uint32_t)kRadix ? gtable[t] : 0u;
        const uint32_t incl = wave_inclusive_sum(total);
        if (lane == 63 && w < (uint32_t)DWAVES) s_wsum[w] = incl;
        __syncthreads();
        if (t == b) {
            uint32_t wave_prefix = 0;
            for (uint32_t i = 0; i < w; i++) wave_prefix += s_wsum[i];
            s_start = incl - total + wave_prefix;
            s_size = total;
        }
        __syncthreads();
    }
    const uint32_t start = s_start, size = s_size;
    LBVH_BT(1);
#ifdef LBVH_BUCKET_TIMING
    if (threadIdx.x == 0) g_bucket_timing[blockIdx.x][39] = size;
#endif
    if (size == 0) return;
    // The bucket covers the fine bins [first, last]: its keys lie in [first << fine_shift, ((last + 1) << fine_shift) - 1] — the last
    // bin holds everything from 4095 << fine_shift up (0xFFFFFFFF pads beside Morton codes).  Sorted by key - base, the bytes of
    // that span are all the passes it takes; the keys travel as differences and get their base back on the way out.
    uint32_t base;
    const uint32_t passes = bucket_passes(gtable[kRadix + b], gtable[2 * kRadix + b], fine_shift, base);
    uint32_t key[ITEMS], lpos2[ITEMS / 2];
    uint32_t digit_total;

    if (size <= (uint32_t)TILE) {
        // ---- the whole bucket in this workgroup's registers -------------------------------------------------------------------
        // LBVH_BUCKET_WAVE_PAIRS pairs per active wave (6 items) until all sixteen waves are in use, then more items per wave
        uint32_t val[ITEMS];
        const uint32_t aw = min((size + (uint32_t)LBVH_BUCKET_WAVE_PAIRS - 1u) / (uint32_t)LBVH_BUCKET_WAVE_PAIRS, (uint32_t)WAVES);
        const uint32_t it = (size + aw * LBVH_WAVE - 1u) / (aw * LBVH_WAVE);          // 1 .. ITEMS
        const uint32_t my_it = w < aw ? it : 0u;
        const uint32_t stripe = it * LBVH_WAVE;
        // (buffer loads: a 32-bit byte offset per load instead of a 64-bit address — thirty-two loads in flight inside the
        // register budget; lanes past the bucket's end read 0 and become the largest key)
        const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(keys_in + start, 0, (int)(size * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(vals_in + start, 0, (int)(size * 4u), 0x00020000);
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            if ((uint32_t)i >= my_it) continue;
            const uint32_t idx = w * stripe + (uint32_t)i * LBVH_WAVE + lane;
            const uint32_t k = __builtin_amdgcn_raw_buffer_load_b32(k_rsrc, idx * 4u, 0, 0);
            key[i] = idx < size ? k - base : 0xFFFFFFFFu;
            val[i] = __builtin_amdgcn_raw_buffer_load_b32(v_rsrc, idx * 4u, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LBVH_BT(2);
        for (uint32_t p = 0; p < passes; p++) {
            bucket_rank<THREADS, ITEMS>(key, aw, my_it, 8u * p, s_xchg, &s_cnt[0][0], s_wcnt, s_dstart, s_wsum, lpos2, digit_total);
            __syncthreads();
            LBVH_BT(3 + 4 * p);
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                if ((uint32_t)i >= my_it) continue;
                s_pair[(i & 1) ? lpos2[i / 2] >> 16 : lpos2[i / 2] & 0xFFFFu] = make_uint2(key[i], val[i]);
            }
            __syncthreads();
            LBVH_BT(4 + 4 * p);
            if (p + 1 == passes) {
                for (uint32_t pos = t; pos < size; pos += THREADS) {
                    const uint2 kv = s_pair[pos];
                    keys_out[start + pos] = kv.x + base;
                    vals_out[start + pos] = kv.y;
                }
                LBVH_BT(30);
                return;
            }
#pragma unroll
            for (int i = 0; i < ITEMS; i++) {
                if ((uint32_t)i >= my_it) continue;
                const uint2 kv = s_pair[w * stripe + (uint32_t)i * LBVH_WAVE + lane];
                key[i] = kv.x;
                val[i] = kv.y;
            }
            __syncthreads();
            LBVH_BT(5 + 4 * p);
        }
        return;
    }

    // ---- a bucket beyond the registers of one workgroup: LSD passes through global memory, chunk by chunk -----------------------
    // Correct for any input and slow (one workgroup moves the whole bucket `passes` times): the form is only chosen when the
    // last sort had no such bucket.  Ping-pong between the two buffers' [start, start + size) regions.
    // (8 pairs per thread and chunk here: this path must not cost the other one registers)
    constexpr int SI = ITEMS < 8 ? ITEMS : 8, STILE = THREADS * SI;
    uint32_t skey[SI], slpos2[SI / 2];
    uint32_t *src_k = keys_in, *src_v = vals_in, *dst_k = keys_out, *dst_v = vals_out;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t pshift = 8u * p;
        if (t < (uint32_t)kRadix) s_base[t] = 0;
        if (t == 0) s_same = 0u;
        __syncthreads();
        {   // (runs of equal digits are added as one: a bucket of ONE key would otherwise put 2 M atomics on one LDS word per pass)
            uint32_t run_digit = 0, run = 0;
            for (uint32_t i0 = 0; i0 < size; i0 += 8u * THREADS) {      // eight loads in flight per thread
                uint32_t k8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t idx = i0 + (uint32_t)u * THREADS + t;
                    k8[u] = idx < size ? src_k[start + idx] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (i0 + (uint32_t)u * THREADS + t >= size) continue;
                    const uint32_t dgt = ((k8[u] - base) >> pshift) & (kRadix - 1);
                    if (dgt != run_digit && run != 0) { atomicAdd(&s_base[run_digit], run); run = 0; }
                    run_digit = dgt;
                    run++;
                }
            }
            if (run != 0) atomicAdd(&s_base[run_digit], run);
        }
        __syncthreads();
        // every pair has the same digit: the pass is the identity (the whole bucket one key — ten thousand triangles in one Morton
        // cell — costs its passes' counting walks, nothing else: 2^21 - 1 equal keys 5.4 -> 0.3 ms, profiles/r6/sort_cliff.txt)
        if (t < (uint32_t)kRadix && s_base[t] == size) s_same = 1u;
        __syncthreads();
        const uint32_t same = s_same;
        __syncthreads();                       // (the next pass clears the word)
        if (same) continue;
        {   // exclusive scan of the digit counts in place
            const uint32_t total = t < (uint32_t)kRadix ? s_base[t] : 0u;
            const uint32_t incl = wave_inclusive_sum(total);
            if (lane == 63 && w < (uint32_t)DWAVES) s_wsum[w] = incl;
            __syncthreads();
            if (t < (uint32_t)kRadix) {
                uint32_t wave_prefix = 0;
#pragma unroll
                for (int i = 0; i < DWAVES; i++) wave_prefix += (uint32_t)i < w ? s_wsum[i] : 0u;
                s_base[t] = incl - total + wave_prefix;
            }
            __syncthreads();
        }
        for (uint32_t c0 = 0; c0 < size; c0 += (uint32_t)STILE) {
            const uint32_t nvalid = min((uint32_t)STILE, size - c0);
#pragma unroll
            for (int i = 0; i < SI; i++) {
                const uint32_t idx = w * (uint32_t)(LBVH_WAVE * SI) + (uint32_t)i * LBVH_WAVE + lane;
                skey[i] = idx < nvalid ? src_k[start + c0 + idx] - base : 0xFFFFFFFFu;
            }
            bucket_rank<THREADS, SI>(skey, (uint32_t)WAVES, (uint32_t)SI, pshift, s_xchg, &s_cnt[0][0], s_wcnt, s_dstart, s_wsum, slpos2, digit_total);
#pragma unroll
            for (int i = 0; i < SI; i++) {
                const uint32_t idx = w * (uint32_t)(LBVH_WAVE * SI) + (uint32_t)i * LBVH_WAVE + lane;
                const uint32_t d = (skey[i] >> pshift) & (kRadix - 1);
                const uint32_t lpos = (i & 1) ? slpos2[i / 2] >> 16 : slpos2[i / 2] & 0xFFFFu;
                if (idx < nvalid) {
                    const uint32_t dst = start + s_base[d] + (lpos - s_dstart[d]);        // its rank among the chunk's pairs of digit d
                    dst_k[dst] = skey[i] + base;
                    dst_v[dst] = src_v[start + c0 + idx];
                }
            }
            __syncthreads();
            if (t < (uint32_t)kRadix) {
                // (the chunk's padding slots were counted under digit 255: they are not pairs)
                s_base[t] += digit_total - (t == (uint32_t)kRadix - 1u ? (uint32_t)STILE - nvalid : 0u);
            }
            __syncthreads();
        }
        // what this workgroup wrote is what it reads next: through the coherence point (rare path: the fence's cost is not an issue)
        __threadfence();
        __syncthreads();
        uint32_t* tmp;
        tmp = src_k; src_k = dst_k; dst_k = tmp;
        tmp = src_v; src_v = dst_v; dst_v = tmp;
    }
    if (src_k != keys_out) {          // an even number of passes (the last bucket's four; none at all: one key) ends in the other buffer
        for (uint32_t i0 = 0; i0 < size; i0 += 8u * THREADS) {          // (eight pairs in flight per thread)
            uint32_t k8[8], v8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t idx = i0 + (uint32_t)u * THREADS + t;
                if (idx < size) { k8[u] = src_k[start + idx]; v8[u] = src_v[start + idx]; }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t idx = i0 + (uint32_t)u * THREADS + t;
                if (idx < size) { keys_out[start + idx] = k8[u]; vals_out[start + idx] = v8[u]; }
            }
        }
    }
}

template <int THREADS, int ITEMS, bool STREAM>
void launch_passes(lbvh_context* ctx, uint32_t* d_keys, uint32_t* d_values, uint32_t* alt_keys, uint32_t* alt_vals,
                   uint32_t count, uint32_t tiles, uint32_t* ghist, uint32_t* status, uint32_t* gstatus, uint32_t groups,
                   uint32_t* tickets, uint32_t group, const uint32_t* gfine = nullptr, uint32_t* bucket_stat = nullptr,
                   uint32_t fine_shift = 0u, uint32_t fine_mul = 0u)
{
    uint32_t *ks = d_keys, *vs = d_values, *kd = alt_keys, *vd = alt_vals;
    for (uint32_t p = 0; p < (uint32_t)kPasses; p++) {   // ComputeBufferSorter.cs:102
        LBVH_LAUNCH(ctx, (sort_onesweep_kernel<THREADS, ITEMS, STREAM>), dim3(tiles), dim3(THREADS), ks, vs, kd, vd, count,
                    8u * p, ghist + p * kRadix, p == 0 ? gfine : nullptr, bucket_stat, fine_shift, fine_mul, (uint32_t*)nullptr,
                    status + (size_t)p * tiles * kRadix,
                    gstatus + (size_t)p * groups * kRadix, tickets + 8u * p, tiles, group, ctx->sort_queues, ctx->fault_dev);
        uint32_t* tmp;
        tmp = ks; ks = kd; kd = tmp;
        tmp = vs; vs = vd; vd = tmp;
    }
}

}  // namespace

namespace {
struct sort_plan {
    int items;
    uint32_t tiles, groups;
    uint32_t *alt_keys, *alt_vals, *ghist, *tickets, *status, *gstatus;
    size_t zero_bytes;
};
}

static int sort_prepare(lbvh_context* ctx, uint32_t count, sort_plan* pl)
{
    // tile = 512 threads x 16 keys (8192) for big inputs: long digit runs = fuller cache lines in the
    // scatter and short look-back chains (measured best of 256..1024 threads x 4..16 keys at 2^24..2^28);
    // 512 x 8 below 2 M keys so every CU still gets tiles
    const int threads = 512, items = count >= (1u << 21) ? 16 : 8;
    const uint32_t tile = (uint32_t)threads * (uint32_t)items;
    const uint32_t tiles = (uint32_t)(((uint64_t)count + tile - 1) / tile);
    const size_t pair_bytes = (((size_t)count * 4) + 255) & ~(size_t)255;
    // [ghist 4x256 | the two-level form's 4096 fine bins | its bucket table 3x256 | tickets (4 passes x 8 XCDs, padded to 256 B) |
    // tile words 4 x tiles x 256 | group words] is zeroed per sort
    const size_t head_bytes = (size_t)(kPasses * kRadix + kFineBins + 3 * kRadix) * 4 + 256;
    const uint32_t groups = (tiles + (uint32_t)kLbGroup - 1u) / (uint32_t)kLbGroup;
    const size_t status_bytes = (size_t)kPasses * ((size_t)tiles + groups) * kRadix * 4;
    int rc = lbvh_reserve(ctx, &ctx->sort_scratch, &ctx->sort_scratch_bytes, 2 * pair_bytes + head_bytes + status_bytes);
    if (rc != LBVH_OK) return rc;
    char* p = (char*)ctx->sort_scratch;
    pl->items = items;
    pl->tiles = tiles;
    pl->groups = groups;
    pl->alt_keys = (uint32_t*)p;
    pl->alt_vals = (uint32_t*)(p + pair_bytes);
    pl->ghist = (uint32_t*)(p + 2 * pair_bytes);
    pl->tickets = pl->ghist + kPasses * kRadix + kFineBins + 3 * kRadix;
    pl->status = (uint32_t*)(p + 2 * pair_bytes + head_bytes);
    pl->gstatus = pl->status + (size_t)kPasses * tiles * kRadix;
    pl->zero_bytes = head_bytes + status_bytes;
    return LBVH_OK;
}

int lbvh_sort_scratch(lbvh_context* ctx, uint32_t count, uint32_t** d_zero, uint32_t* zero_words)
{
    *d_zero = nullptr;
    *zero_words = 0;
    if (count < 2) return LBVH_OK;
    sort_plan pl;
    const int rc = sort_prepare(ctx, count, &pl);
    if (rc != LBVH_OK) return rc;
    if (pl.zero_bytes / 4 > 0xFFFFFFFFull) return LBVH_OK;      // the caller's sort clears it itself
    *d_zero = pl.ghist;
    *zero_words = (uint32_t)(pl.zero_bytes / 4);
    return LBVH_OK;
}

// sizes at which a pass is latency, not bandwidth (below: too few pairs to matter; above: 16 K-key tiles, bandwidth-bound passes —
// there an MSD-first form was measured to gain nothing: profiles/r4/b_bucket_local_sort_passes.txt)
static inline bool two_level_size(uint32_t count) { return count >= (1u << 15) && count < (1u << 21); }
constexpr int kBucketThreads = 1024, kBucketItems = 16;        // a bucket of up to 16 384 pairs lives in one workgroup's registers
constexpr uint32_t kSortStreak = 3;                            // sorts in a row whose largest bucket fitted, before the two-level form is taken

int lbvh_launch_sort(lbvh_context* ctx, uint32_t* d_keys, uint32_t* d_values, uint32_t count, bool scratch_cleared, uint32_t key_bits)
{
    if (count < 2) return LBVH_OK;
    sort_plan pl;
    {
        const int rc = sort_prepare(ctx, count, &pl);
        if (rc != LBVH_OK) return rc;
    }
    const int items = pl.items;
    const uint32_t tiles = pl.tiles, groups = pl.groups;
    uint32_t *alt_keys = pl.alt_keys, *alt_vals = pl.alt_vals, *ghist = pl.ghist, *tickets = pl.tickets, *status = pl.status,
             *gstatus = pl.gstatus;
    if (!scratch_cleared) LBVH_HIP_TRY(ctx, hipMemsetAsync(ghist, 0, pl.zero_bytes, ctx->cur_stream));

    // 4 K keys per block up to 2048 blocks: enough blocks to hide the load latency, few enough that the
    // 1024 global atomics each block ends with do not pile up on the same counters
    uint32_t hblocks = (count + 8191u) / 8192u;
    if (hblocks > 256u * 8u) hblocks = 256u * 8u;
    // consecutive tiles per XCD: 16 when every XCD still gets several groups, fewer for small sorts
    const uint32_t group = tiles >= 1024u ? 16u : tiles >= 128u ? 8u : 1u;
    // The form.  Two-level where a pass is latency (two_level_size) and the LAST sort of this context, bucketed the same way, had
    // no bucket beyond one workgroup's registers — the first pass kernel of either form leaves the largest (balanced) bucket of
    // ITS input in four mapped host words, tagged with the fine shift; read here without any synchronisation: a hint, one sort
    // stale (a first sort, or one after an input with more than ~16 K equal 12-bit prefixes, takes the four passes; such an input
    // after a spread one is sorted correctly by the bucket kernel's slow path, once).  The fine bins are the 12 bits below the
    // caller's key_bits (the rebuild: Morton codes < 2^30 -> bits 18 .. 29, pads in the last bin; lbvh_sort_pairs: the top 12).
    // lbvh_debug_switch(LBVH_DEBUG_SORT_FORM): 1 four passes always, 2 two-level wherever the size allows.
    const uint32_t fine_shift = (key_bits >= (uint32_t)kFineLog2 && key_bits <= 32u ? key_bits : 32u) - (uint32_t)kFineLog2;
    const uint32_t fine_mul = (uint32_t)((1ull << 40) / count);
    uint32_t* gfine = ghist + kPasses * kRadix;
    uint32_t* gtable = gfine + kFineBins;
    uint32_t* stat_dev = ctx->fault_dev + 16;                   // words 16 .. 19 of the mapped block (word 0: the fault word)
    bool two_level = false;
    if (two_level_size(count)) {
        const volatile uint32_t* st = ctx->fault_host + 16;
        uint32_t largest = 0;
        bool known = true;
        for (int w = 0; w < 4; w++) {
            const uint32_t v = st[w];
            known = known && v != 0xFFFFFFFFu && (v >> 24) == fine_shift;
            largest = std::max(largest, v & 0xFFFFFFu);
        }
        // ... and not on ONE such sort: kSortStreak calls in a row must have found the hint in order (round 6).  The bucket kernel
        // sorts a bucket beyond its registers correctly but with ONE workgroup, chunk by chunk through memory — milliseconds for a
        // bucket of a million pairs (profiles/r6/sort_cliff.txt) against ~60 us for the four passes.  With the streak an input that
        // alternates between spread and degenerate (a scene collapsing into one Morton cell every other frame) stays in
        // the four-pass form; what is left is one slow sort at the first degenerate input after at least kSortStreak spread ones.
        const bool in_order = known && largest <= (uint32_t)(kBucketThreads * kBucketItems);
        ctx->sort_hint_streak = in_order ? std::min(ctx->sort_hint_streak + 1u, 1u << 20) : 0u;
        const uint32_t form = ctx->debug_switch[LBVH_DEBUG_SORT_FORM];
        two_level = form == 2u || (form == 0u && ctx->sort_hint_streak >= kSortStreak);
        // a chain being captured into a graph is replayed for inputs this call knows nothing about: a frozen hint is no hint
        //  — captured sorts take the input-independent four passes
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (form != 2u && hipStreamIsCapturing(ctx->cur_stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone)
            two_level = false;
    }
    if (two_level) {
        const uint32_t fblocks = std::max(1u, std::min(hblocks, (count + (uint32_t)LBVH_FINE_BLOCK_KEYS - 1u) / (uint32_t)LBVH_FINE_BLOCK_KEYS));
        LBVH_LAUNCH(ctx, sort_histogram_kernel<HIST_FINE>, dim3(fblocks), dim3(kThreads), d_keys, count, ghist, gfine, fine_shift);
        LBVH_LAUNCH(ctx, (sort_onesweep_kernel<512, 8, false, true>), dim3(tiles), dim3(512), d_keys, d_values, alt_keys, alt_vals, count,
                    fine_shift, ghist, gfine, stat_dev, fine_shift, fine_mul, gtable, status, gstatus, tickets, tiles, group, ctx->sort_queues,
                    ctx->fault_dev);
        LBVH_LAUNCH(ctx, (sort_bucket_kernel<kBucketThreads, kBucketItems>), dim3(kRadix), dim3(kBucketThreads), alt_keys, alt_vals, d_keys,
                    d_values, fine_shift, gtable);
        LBVH_HIP_TRY(ctx, hipGetLastError());
#ifdef LBVH_BUCKET_TIMING
        {
            static unsigned long long h[kRadix][40];
            (void)hipStreamSynchronize(ctx->cur_stream);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bucket_timing), sizeof h);
            int big = 0;
            for (int b2 = 0; b2 < kRadix; b2++) if (h[b2][39] > h[big][39] && h[b2][39] <= 16384) big = b2;
            fprintf(stderr, "[bucket timing] largest bucket %d (%llu pairs), s_memtime ticks since the kernel's first instruction:", big, h[big][39]);
            const int slots[] = {1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 30, 32, 33, 34};
            for (int q : slots) fprintf(stderr, " [%d] %lld", q, (long long)(h[big][q] - h[big][0]));
            fprintf(stderr, "\n");
        }
#endif
        return LBVH_OK;       // the buckets are back in d_keys / d_values
    }
    const bool stat = two_level_size(count);      // (then items == 8: the 512 x 8 form, whose first tile can derive the buckets)
    if (stat)
        LBVH_LAUNCH(ctx, (sort_histogram_kernel<HIST_FOUR | HIST_FINE>), dim3(std::max(1u, std::min(hblocks, (count + (uint32_t)LBVH_FINE_BLOCK_KEYS - 1u) / (uint32_t)LBVH_FINE_BLOCK_KEYS))),
                    dim3(kThreads), d_keys, count, ghist, gfine, fine_shift);
    else
        LBVH_LAUNCH(ctx, sort_histogram_kernel<HIST_FOUR>, dim3(hblocks), dim3(kThreads), d_keys, count, ghist, gfine, fine_shift);
    if (items == 16 && count >= (1u << 23))
        launch_passes<512, 16, true>(ctx, d_keys, d_values, alt_keys, alt_vals, count, tiles, ghist, status, gstatus, groups, tickets, group);
    else if (items == 16)
        launch_passes<512, 16, false>(ctx, d_keys, d_values, alt_keys, alt_vals, count, tiles, ghist, status, gstatus, groups, tickets, group);
    else
        launch_passes<512, 8, false>(ctx, d_keys, d_values, alt_keys, alt_vals, count, tiles, ghist, status, gstatus, groups, tickets, group,
                                     stat ? gfine : nullptr, stat_dev, fine_shift, fine_mul);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;   // 4 passes: the result is back in d_keys / d_values
}

extern "C" lbvh_status lbvh_sort_pairs(lbvh_context* ctx, uint32_t* d_keys, uint32_t* d_values,
                                       uint32_t count)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (count == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_keys != nullptr && d_values != nullptr);
    LBVH_REQUIRE(ctx, count <= kValueMask);       // status words carry 30-bit counts
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    lbvh_note_write(ctx, d_keys, (size_t)count * 4);
    lbvh_note_write(ctx, d_values, (size_t)count * 4);
    return (lbvh_status)lbvh_launch_sort(ctx, d_keys, d_values, count, false, 32u);
}


// Host-side view of the tile order (no GPU involved): the tile that ticket k of queue x stands for.  For the CPU
// model test of the order invariant (tests/test_abi.py): every tile below a handed-out tile is handed out already
// or is the next ticket of some queue.
extern "C" uint32_t lbvh_debug_sort_ticket_tile(uint32_t k, uint32_t x, uint32_t group, uint32_t queues)
{
    return ticket_tile(k, x, group, queues);
}

// ---- cfg4: local kernels of the key-range sharded sort ---------------------------------------------

struct probe_args { uint32_t v[64]; };
struct prefix_args { uint32_t v[16]; };

// one 8-bit digit histogram per selected key prefix, privatised in LDS
// (prefixes: by value from the host, or — d_prefixes != nullptr — read from device memory: the sharded sort keeps its
// splitter search on the device between the all-reduces)
__global__ __launch_bounds__(256) void key_histogram_kernel(const uint32_t* __restrict__ keys, uint32_t count,
                                                            prefix_args prefixes, const uint32_t* __restrict__ d_prefixes,
                                                            uint32_t n_prefixes, uint32_t prefix_shift, uint32_t shift,
                                                            uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_hist[16 * 256];
    const uint32_t bins = n_prefixes * 256u;
    for (uint32_t i = threadIdx.x; i < bins; i += 256) s_hist[i] = 0;
    if (d_prefixes)
        for (uint32_t p = 0; p < n_prefixes; ++p) prefixes.v[p] = d_prefixes[p];
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const uint32_t k = keys[i];
        const uint32_t d = (k >> shift) & 255u;
        if (prefix_shift >= 32u) {
            atomicAdd(&s_hist[d], 1u);
        } else {
            const uint32_t top = k >> prefix_shift;
            for (uint32_t p = 0; p < n_prefixes; ++p)
                if (top == prefixes.v[p]) atomicAdd(&s_hist[p * 256u + d], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += 256) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

__global__ void lower_bound_kernel(const uint32_t* __restrict__ keys, uint32_t count, probe_args probes,
                                   const uint32_t* __restrict__ d_probes, uint32_t n_probes, uint32_t* __restrict__ out)
{
    const uint32_t j = threadIdx.x;
    if (j >= n_probes) return;
    const uint32_t probe = d_probes ? d_probes[j] : probes.v[j];
    uint32_t lo = 0, hi = count;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < probe) lo = mid + 1; else hi = mid;
    }
    out[j] = lo;
}

static lbvh_status key_histogram_impl(lbvh_context* ctx, const uint32_t* d_keys, uint32_t count, const uint32_t* h_prefixes,
                                      const uint32_t* d_prefixes, uint32_t n_prefixes, uint32_t prefix_shift, uint32_t shift,
                                      uint32_t* d_hist)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, d_hist != nullptr && n_prefixes >= 1 && n_prefixes <= 16 && shift <= 24 && prefix_shift <= 32);
    LBVH_REQUIRE(ctx, (prefix_shift == 32 && n_prefixes == 1) || (prefix_shift < 32 && (h_prefixes != nullptr || d_prefixes != nullptr)));
    LBVH_REQUIRE(ctx, count == 0 || d_keys != nullptr);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    LBVH_HIP_TRY(ctx, hipMemsetAsync(d_hist, 0, (size_t)n_prefixes * 256 * 4, ctx->cur_stream));
    if (count == 0) return LBVH_OK;
    prefix_args pa = {};
    if (prefix_shift < 32 && h_prefixes) for (uint32_t p = 0; p < n_prefixes; ++p) pa.v[p] = h_prefixes[p];
    uint32_t blocks = (count + 4095u) / 4096u;
    if (blocks > 2048u) blocks = 2048u;
    LBVH_LAUNCH(ctx, key_histogram_kernel, dim3(blocks), dim3(256), d_keys, count, pa, prefix_shift < 32 ? d_prefixes : nullptr, n_prefixes,
                prefix_shift, shift, d_hist);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

extern "C" lbvh_status lbvh_key_histogram(lbvh_context* ctx, const uint32_t* d_keys, uint32_t count,
                                          const uint32_t* h_prefixes, uint32_t n_prefixes, uint32_t prefix_shift,
                                          uint32_t shift, uint32_t* d_hist)
{
    return key_histogram_impl(ctx, d_keys, count, h_prefixes, nullptr, n_prefixes, prefix_shift, shift, d_hist);
}

extern "C" lbvh_status lbvh_key_histogram_device(lbvh_context* ctx, const uint32_t* d_keys, uint32_t count,
                                                 const uint32_t* d_prefixes, uint32_t n_prefixes, uint32_t prefix_shift,
                                                 uint32_t shift, uint32_t* d_hist)
{
    return key_histogram_impl(ctx, d_keys, count, nullptr, d_prefixes, n_prefixes, prefix_shift, shift, d_hist);
}

static lbvh_status lower_bound_impl(lbvh_context* ctx, const uint32_t* d_sorted_keys, uint32_t count, const uint32_t* h_probes,
                                    const uint32_t* d_probes, uint32_t n_probes, uint32_t* d_positions)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (n_probes == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, (h_probes != nullptr || d_probes != nullptr) && d_positions != nullptr && n_probes <= 64);
    LBVH_REQUIRE(ctx, count == 0 || d_sorted_keys != nullptr);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    probe_args pa = {};
    if (h_probes) for (uint32_t j = 0; j < n_probes; ++j) pa.v[j] = h_probes[j];
    LBVH_LAUNCH(ctx, lower_bound_kernel, dim3(1), dim3(64), d_sorted_keys, count, pa, d_probes, n_probes, d_positions);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

extern "C" lbvh_status lbvh_lower_bound(lbvh_context* ctx, const uint32_t* d_sorted_keys, uint32_t count,
                                        const uint32_t* h_probes, uint32_t n_probes, uint32_t* d_positions)
{
    return lower_bound_impl(ctx, d_sorted_keys, count, h_probes, nullptr, n_probes, d_positions);
}

extern "C" lbvh_status lbvh_lower_bound_device(lbvh_context* ctx, const uint32_t* d_sorted_keys, uint32_t count,
                                               const uint32_t* d_probes, uint32_t n_probes, uint32_t* d_positions)
{
    return lower_bound_impl(ctx, d_sorted_keys, count, nullptr, d_probes, n_probes, d_positions);
}
