// lbvh_build.hip — Morton/AABB generation, DistributeKeys, Karras topology, AABB refit (gfx950).
//
// Reference: Assets/_Scripts/MeshBufferContainer.cs:32-83,123-146 (CPU loop -> kernel here),
// :154-169 (CPU DistributeKeys -> 3-kernel scan here), Assets/_Shaders/BVH/BVH.compute:18-149
// (TreeConstructor) and :152-220 (BVHConstructor).  All fp32 arithmetic is strict (the library is
// compiled with -ffp-contract=off) so results are bit-identical to the CPU oracle.
#include "lbvh_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// a-1  Morton codes + triangle AABBs
// ---------------------------------------------------------------------------------------------
struct box3 { float mn[3]; float mx[3]; };

__device__ __forceinline__ uint32_t expand_bits(uint32_t v)   // MeshBufferContainer.cs:32-39
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__device__ __forceinline__ uint32_t quantize(float x)          // MeshBufferContainer.cs:43-48
{
    x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    return (uint32_t)x;
}

// LBVH_BUILD_RESET_NODES without two fills of 32 MB in front of the chain: the tree kernel that follows writes EVERY word of
// internal nodes [0, n - 1) and leaves [0, n) — a node's own five words by its thread, its parent word by its parent's —
// except the root's parent word, so refilling the node arrays with 0xFFFFFFFF (Sc/MeshBufferContainer.cs:114-115) comes down
// to the slots past the tree and that one word: thread i of the Morton kernel does slot i.
struct node_reset { uint32_t* internal; uint32_t* leaf; };      // nullptr: no reset
__device__ __forceinline__ void reset_node_slot(const node_reset& r, uint32_t i, uint32_t n, uint32_t capacity)
{
    if (!r.internal || i >= capacity) return;
    if (i + 1u >= n) {          // internal slots n - 1 .. capacity - 1
#pragma unroll
        for (int k = 0; k < 6; k++) r.internal[(size_t)i * 6u + k] = 0xFFFFFFFFu;
    }
    if (i >= n) { r.leaf[(size_t)i * 2u] = 0xFFFFFFFFu; r.leaf[(size_t)i * 2u + 1u] = 0xFFFFFFFFu; }
    if (i == 0u) r.internal[4] = 0xFFFFFFFFu;          // the root's parent: never written by TreeConstructor (BVH.compute:94-149)
}

// one triangle's Morton code, index, padded AABB (and, for lbvh_build_scene, its 64-byte traversal line) from its three
// positions: the loop body of MeshBufferContainer.cs:123-146.  Key and index go to memory, box and line to the workgroup's
// LDS staging (the caller stores them as whole records).
__device__ __forceinline__ void morton_of_triangle(const float4 a, const float4 b, const float4 c, uint32_t i, const box3& scene,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ indices, float4* s_box,
                                                   float4* s_line, bool with_line)
{
    const float ax[3] = {a.x, a.y, a.z}, bx[3] = {b.x, b.y, b.z}, cx[3] = {c.x, c.y, c.z};
    float mn[3], mx[3];
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        mn[k] = fminf(fminf(ax[k], bx[k]), cx[k]) - 0.001f;      // GetCentroidAndAABB :54-63
        mx[k] = fmaxf(fmaxf(ax[k], bx[k]), cx[k]) + 0.001f;
        float cen = (mn[k] + mx[k]) * 0.5f;                      // :65
        cen = cen - scene.mn[k];                                 // NormalizeCentroid :76-81
        cen = cen / (scene.mx[k] - scene.mn[k]);
        q[k] = quantize(cen);
    }
    keys[i] = expand_bits(q[0]) * 4u + expand_bits(q[1]) * 2u + expand_bits(q[2]);   // :46-49
    indices[i] = i;
    s_box[threadIdx.x * 2 + 0] = make_float4(mn[0], mn[1], mn[2], 0.0f);
    s_box[threadIdx.x * 2 + 1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
    if (with_line) {
        // lbvh_build_scene: the derived scene's triangle line (lbvh_common.h lbvh_fast_tri: first vertex, the two edge
        // vectors of Raytracing.compute:41-42, the triangle's index), in ORIGINAL order — the positions are in
        // registers here, so no kernel has to gather the 128-byte records again after the sort
        const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
        const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
        s_line[threadIdx.x * 4 + 0] = make_float4(a.x, a.y, a.z, __uint_as_float(i));
        s_line[threadIdx.x * 4 + 1] = make_float4(a.x, a.y, a.z, e2x);
        s_line[threadIdx.x * 4 + 2] = make_float4(e1x, e1y, e1z, e2y);
        s_line[threadIdx.x * 4 + 3] = make_float4(e1x, e1y, e1z, e2z);
    }
}

__global__ __launch_bounds__(256) void morton_aabb_kernel(const lbvh_triangle* __restrict__ tris,
                                                          uint32_t n, uint32_t capacity, box3 scene,
                                                          uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ indices,
                                                          lbvh_aabb* __restrict__ aabb,
                                                          uint32_t* __restrict__ zero, uint32_t zero_words,
                                                          lbvh_fast_tri* __restrict__ lines, node_reset reset)
{
    const uint32_t b0 = blockIdx.x * 256u, i = b0 + threadIdx.x;
    // lbvh_build_scene: the sort that follows wants its counters and look-back words cleared; doing it here saves
    // a fill kernel in the chain
    for (uint32_t w = i; w < zero_words; w += gridDim.x * 256u) zero[w] = 0u;
    reset_node_slot(reset, i, n, capacity);
    // records leave through LDS: a thread produces one triangle's 32-byte AABB and 64-byte line, the workgroup stores
    // them as consecutive float4 (a wave's store = 1 KB of whole records instead of 64 half / quarter lines)
    __shared__ float4 s_box[256 * 2];
    __shared__ float4 s_line[256 * 4];
    if (i < capacity && i >= n) {   // DataBuffer<uint>(.., uint.MaxValue)  MeshBufferContainer.cs:108-109
        keys[i] = 0xFFFFFFFFu;
        indices[i] = 0xFFFFFFFFu;
    }
    if (i < n) {
        // only the three padded positions (48 of the 128 bytes) are read
        const float4* p = reinterpret_cast<const float4*>(&tris[i]);
        morton_of_triangle(p[0], p[1], p[2], i, scene, keys, indices, s_box, s_line, lines != nullptr);
    }
    __syncthreads();
    const uint32_t live = n > b0 ? min(n - b0, 256u) : 0u;          // triangles of this block
    float4* ob = reinterpret_cast<float4*>(aabb + b0);
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) {
        const uint32_t f = k * 256u + threadIdx.x;
        if (f < live * 2u) ob[f] = s_box[f];
    }
    if (lines) {
        float4* ol = reinterpret_cast<float4*>(lines + b0);
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t f = k * 256u + threadIdx.x;
            if (f < live * 4u) ol[f] = s_line[f];
        }
    }
}

// lbvh_animate + the kernel above in one pass (lbvh_animate_build_scene: the per-frame chain of a dynamic scene).  As two
// kernels the moved triangles were written (128 B each) and read straight back for the 36 bytes of their positions: 128 MB
// per million triangles fetched for nothing.  Here a workgroup moves its 256 triangles — one thread per 16-byte quarter row,
// loads and stores of a wave covering 1 KB of consecutive bytes, as animate_kernel — keeps the moved positions in LDS, and
// every thread then takes one triangle's Morton code, box and line from there.  Same floats in the same order as the two
// kernels: the results are identical word for word.
__global__ __launch_bounds__(256) void animate_morton_kernel(const lbvh_triangle* __restrict__ rest, const uint32_t* __restrict__ body,
                                                             const float4* __restrict__ centres, float cs, float sn,
                                                             lbvh_triangle* __restrict__ tris, uint32_t n, uint32_t capacity, box3 scene,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ indices,
                                                             lbvh_aabb* __restrict__ aabb, uint32_t* __restrict__ zero, uint32_t zero_words,
                                                             lbvh_fast_tri* __restrict__ lines, node_reset reset)
{
    const uint32_t b0 = blockIdx.x * 256u, i = b0 + threadIdx.x;
    for (uint32_t w = i; w < zero_words; w += gridDim.x * 256u) zero[w] = 0u;
    reset_node_slot(reset, i, n, capacity);
    __shared__ float4 s_pos[256 * 3];
    __shared__ float4 s_box[256 * 2];
    __shared__ float4 s_line[256 * 4];
    if (i < capacity && i >= n) {
        keys[i] = 0xFFFFFFFFu;
        indices[i] = 0xFFFFFFFFu;
    }
    // quarter row k of the triangles (j * 32 + t / 8), j = 0 .. 7: k = t & 7 is the same for all eight (256 % 8 == 0)
    const uint32_t k = threadIdx.x & 7u;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
        const uint32_t local = j * 32u + (threadIdx.x >> 3), tri = b0 + local;
        if (tri < n) {
            const size_t g = (size_t)tri * 8u + k;
            float4 p = reinterpret_cast<const float4*>(rest)[g];
            if (k < 3u) {                                        // positions a, b, c
                const float4 ctr = centres[body[tri]];
                const float x = p.x - ctr.x, z = p.z - ctr.z;
                p.x = (cs * x + sn * z) + ctr.x;                 // rotation about Y through the body centre (animate_kernel)
                p.z = (cs * z - sn * x) + ctr.z;
                s_pos[local * 3u + k] = p;
            } else if (k >= 5u) {                                // normals (k = 3, 4: uv, copied)
                const float nx = p.x, nz = p.z;
                p.x = cs * nx + sn * nz;
                p.z = cs * nz - sn * nx;
            }
            reinterpret_cast<float4*>(tris)[g] = p;
        }
    }
    __syncthreads();
    if (i < n)
        morton_of_triangle(s_pos[threadIdx.x * 3 + 0], s_pos[threadIdx.x * 3 + 1], s_pos[threadIdx.x * 3 + 2], i, scene, keys, indices, s_box,
                           s_line, lines != nullptr);
    __syncthreads();
    const uint32_t live = n > b0 ? min(n - b0, 256u) : 0u;
    float4* ob = reinterpret_cast<float4*>(aabb + b0);
#pragma unroll
    for (uint32_t q = 0; q < 2; q++) {
        const uint32_t f = q * 256u + threadIdx.x;
        if (f < live * 2u) ob[f] = s_box[f];
    }
    if (lines) {
        float4* ol = reinterpret_cast<float4*>(lines + b0);
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const uint32_t f = q * 256u + threadIdx.x;
            if (f < live * 4u) ol[f] = s_line[f];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// a-6  DistributeKeys: new[i] = sum_{j=1..i} max(old[j] - old[j-1], 1), new[0] = 0
// reduce per chunk -> scan of chunk sums -> apply.  Chunk = 256 threads x 8 consecutive keys.
// ---------------------------------------------------------------------------------------------
constexpr int kDistThreads = 256;
constexpr int kDistItems = 8;
constexpr int kDistChunk = kDistThreads * kDistItems;

__device__ __forceinline__ uint32_t dist_step(uint32_t cur, uint32_t prev)
{
    const uint32_t diff = cur - prev;     // uint arithmetic, MeshBufferContainer.cs:163
    return diff > 1u ? diff : 1u;         // Math.Max(diff, 1)
}

// loads this thread's 8 keys and the key before them; returns the 8 step values f[]
__device__ __forceinline__ void dist_load(const uint32_t* __restrict__ keys, uint32_t n, uint32_t j0,
                                          uint32_t prev, uint32_t f[kDistItems])
{
#pragma unroll
    for (int c = 0; c < kDistItems; c++) {
        const uint32_t j = j0 + (uint32_t)c;
        uint32_t cur = 0;
        if (j < n) cur = keys[j];
        f[c] = (j < n && j > 0) ? dist_step(cur, prev) : 0u;
        prev = cur;
    }
}

__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t* s_wave, uint32_t* total)
{
    const uint32_t t = threadIdx.x;
    const uint32_t incl = wave_inclusive_sum(v);
    if ((t & 63) == 63) s_wave[t >> 6] = incl;
    __syncthreads();
    uint32_t prefix = 0, all = 0;
#pragma unroll
    for (uint32_t i = 0; i < kDistThreads / LBVH_WAVE; i++) {
        const uint32_t x = s_wave[i];
        prefix += i < (t >> 6) ? x : 0u;
        all += x;
    }
    *total = all;
    return incl - v + prefix;
}

// (kernel bodies take their workgroup's index as an argument: lbvh_build_scene runs two of them side by side in ONE launch —
// "merged launches" further down)
__device__ __forceinline__ void distribute_reduce_body(uint32_t block, const uint32_t* __restrict__ keys, uint32_t n,
                                                       uint32_t* __restrict__ chunk_sums, uint32_t* __restrict__ boundary)
{
    __shared__ uint32_t s_wave[kDistThreads / LBVH_WAVE];
    const uint32_t base = block * (uint32_t)kDistChunk;
    const uint32_t j0 = base + threadIdx.x * (uint32_t)kDistItems;
    const uint32_t prev = (j0 > 0 && j0 - 1 < n) ? keys[j0 - 1] : 0u;
    uint32_t f[kDistItems];
    dist_load(keys, n, j0, prev, f);
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < kDistItems; c++) s += f[c];
    uint32_t total;
    (void)block_exclusive_sum(s, s_wave, &total);
    if (threadIdx.x == 0) {
        chunk_sums[block] = total;
        boundary[block] = prev;        // old key just before this chunk: the apply pass must not
    }                                   // re-read it, the previous chunk overwrites it in place
}

__global__ __launch_bounds__(kDistThreads) void distribute_reduce_kernel(
    const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ chunk_sums,
    uint32_t* __restrict__ boundary)
{
    distribute_reduce_body(blockIdx.x, keys, n, chunk_sums, boundary);
}

__global__ __launch_bounds__(kDistThreads) void distribute_scan_kernel(uint32_t* __restrict__ chunk_sums,
                                                                       uint32_t chunks)
{
    __shared__ uint32_t s_wave[kDistThreads / LBVH_WAVE];
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < chunks; c0 += kDistThreads) {
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t x = c < chunks ? chunk_sums[c] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_sum(x, s_wave, &total);
        if (c < chunks) chunk_sums[c] = carry + excl;
        carry += total;
        __syncthreads();
    }
}

// SELF: chunk_sums still holds the raw per-chunk sums (no scan kernel ran): the block adds up its predecessors' itself —
// a few loads per thread up to kSelfScanChunks chunks, and one dependent single-workgroup kernel less in the chain
constexpr uint32_t kSelfScanChunks = 2048;

template <bool SELF>
__device__ __forceinline__ void distribute_apply_body(uint32_t block, uint32_t* __restrict__ keys, uint32_t n,
                                                      const uint32_t* __restrict__ chunk_sums, const uint32_t* __restrict__ boundary)
{
    __shared__ uint32_t s_wave[kDistThreads / LBVH_WAVE];
    __shared__ uint32_t s_before[kDistThreads / LBVH_WAVE];
    uint32_t before = 0;
    if (SELF) {
        uint32_t part = 0;
        for (uint32_t c = threadIdx.x; c < block; c += kDistThreads) part += chunk_sums[c];
        (void)block_exclusive_sum(part, s_before, &before);
    } else {
        before = chunk_sums[block];
    }
    const uint32_t base = block * (uint32_t)kDistChunk;
    const uint32_t j0 = base + threadIdx.x * (uint32_t)kDistItems;
    uint32_t prev = 0;
    if (threadIdx.x == 0) prev = boundary[block];
    else if (j0 - 1 < n) prev = keys[j0 - 1];
    uint32_t f[kDistItems];
    dist_load(keys, n, j0, prev, f);
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < kDistItems; c++) { s += f[c]; f[c] = s; }   // thread-local inclusive
    uint32_t total;
    const uint32_t excl = block_exclusive_sum(s, s_wave, &total);  // barrier inside: every old key
    const uint32_t carry = before + excl;                          // of the chunk is loaded by now
#pragma unroll
    for (int c = 0; c < kDistItems; c++) {
        const uint32_t j = j0 + (uint32_t)c;
        if (j < n) keys[j] = carry + f[c];
    }
}

template <bool SELF>
__global__ __launch_bounds__(kDistThreads) void distribute_apply_kernel(
    uint32_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ chunk_sums,
    const uint32_t* __restrict__ boundary)
{
    distribute_apply_body<SELF>(blockIdx.x, keys, n, chunk_sums, boundary);
}

// ---------------------------------------------------------------------------------------------
// a-7  TreeConstructor (Karras 2012)          BVH.compute:18-149
// a-8  BVHConstructor, fused into it for lbvh_build_scene / lbvh_build_fast_scene (see "range hierarchy" below)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }   // :18-21

// A node's range and split depend on the keys around it: for > 99 % of the nodes on nothing further than a few hundred keys from
// idx.  The workgroup's keys plus a halo on each side are staged in LDS once, coalesced; what a node needs from them is answered
// there — by nearest-set-bit lookups in bitmaps of the adjacent-key prefix array ("search-free form" below; rounds 1 - 5: by the
// reference's probe loops, which remain for windows with equal neighbours) — and a node whose answer lies outside the window is
// handed to its wave ("wide nodes").
constexpr int kTreeThreads = 256;
#ifndef LBVH_TREE_HALO
#define LBVH_TREE_HALO 256               // (measurement builds: 128 / 256 / 384 run 74.5 / 73.4 / 77.8 us at 1 M nodes, round 6)
#endif
constexpr int kTreeHalo = LBVH_TREE_HALO;
constexpr int kTreeWindow = kTreeThreads + 2 * kTreeHalo;

typedef __attribute__((address_space(3))) const uint32_t lds_u32;
typedef unsigned long long u64;
struct key_window {
    lds_u32* lds;                                   // an LDS pointer by TYPE (a generic one makes every probe a flat_load)
    int w0, w1;                                     // keys [w0, w1) are lds[0 .. w1 - w0)
    int num;
    __device__ __forceinline__ bool inside(int y) const { return (uint32_t)(y - w0) < (uint32_t)(w1 - w0); }
    __device__ __forceinline__ uint32_t at(int y) const { return lds[y - w0]; }            // y inside the window
    // delta(x, y) of BVH.compute:23-33 for a y inside the window (x is always in range); `left` is set when y is a
    // valid key OUTSIDE the window: the node is a WIDE one and goes to wide_node_search below
    __device__ __forceinline__ int delta(uint32_t x_code, int y, bool& left) const
    {
        if (y < 0 || y > num - 1) return -1;
        if (!inside(y)) { left = true; return -1; }
        return clz32(x_code ^ at(y));
    }
};

// ---- wide nodes -------------------------------------------------------------------------------------------------------
// A node whose searches leave the LDS window (range beyond ~256 leaves: one node in ~128) would run the outside probes
// as dependent global loads, one after the other while the rest of its wave waits: a handful for most, ~60 for the
// root.  Such a node is handed to its WAVE instead: all 64 lanes probe at once.  Both searches of the reference are
// "last position where a monotone predicate still holds" (delta(idx, idx + l d) is non-increasing in l for strictly
// increasing keys; so is the prefix shared with the first key): any search strategy returns the same position, so a
// 64-ary search (log64 instead of log2 dependent round trips) is bit-identical.
// last l in [lo, hi) with pred(l), given pred(lo) and !pred(hi) (hi may be "one past"): all lanes call, result uniform
template <typename Pred>
__device__ __forceinline__ uint32_t last_true_64ary(uint32_t lo, uint32_t hi, Pred pred)
{
    const uint32_t lane = lane_id();
    while (hi - lo > 1u) {
        const uint32_t span = hi - lo;                               // candidates lo+1 .. hi-1
        const uint32_t step = (span + 63u) / 64u;                    // >= 1; 63 probes lo + step, lo + 2 step, ...
        const uint32_t l = lo + (lane + 1u) * step;
        const bool p = lane < 63u && l < hi && pred(l);
        const uint64_t m = __ballot(p);
        const uint32_t cnt = (uint32_t)__builtin_ctzll(~m);          // consecutive true from lane 0 (lane 63 never is)
        lo += cnt * step;
        hi = min(hi, lo + step);
    }
    return lo;
}

// DetermineRange + FindSplit of node idx by the whole wave (uniform arguments and results).  self, d, dmin: what the node's own
// lane found in the LDS window (BVH.compute:36-38) — one round trip to memory less.
__device__ __forceinline__ void wide_node_range(const uint32_t* __restrict__ codes, int num, int idx, uint32_t self, int d, int dmin,
                                                int& first, int& last)
{
    const uint32_t lane = lane_id();
    auto dlt = [&](int y) -> int { return (y >= 0 && y <= num - 1) ? clz32(self ^ codes[y]) : -1; };      // :23-33
    // :39-41  lmax = 2; while (delta(idx + lmax d) > dmin) lmax *= 2: lane k tries 2 << k, the first failure ends it
    uint32_t lmax;
    {
        const uint32_t cand = 2u << (lane & 31u);
        const bool p = lane < 31u && dlt(idx + (int)(cand * (uint32_t)d)) > dmin;
        const uint64_t m = __ballot(p);
        lmax = 2u << (uint32_t)__builtin_ctzll(~m);
    }
    // :42-47  the largest l < lmax with delta(idx + l d) > dmin (l = 0 always qualifies: delta(idx, idx) = 32)
    const uint32_t l = last_true_64ary(0u, lmax, [&](uint32_t q) { return dlt(idx + (int)(q * (uint32_t)d)) > dmin; });
    const int j = idx + (int)l * d;                                                        // :49
    first = min(idx, j);                                                                   // :50
    last = max(idx, j);
}

// FindSplit :54-92: the last position in [first, last) that shares more than the range's common prefix with `first`
__device__ __forceinline__ int wide_node_split(const uint32_t* __restrict__ codes, int first, int last, uint32_t first_code,
                                               uint32_t last_code)
{
    if (first_code == last_code) return (first + last) >> 1;                               // :61-62
    const int common_prefix = clz32(first_code ^ last_code);                               // :67
    return (int)last_true_64ary((uint32_t)first, (uint32_t)last,
                                [&](uint32_t q) { return clz32(first_code ^ codes[q]) > common_prefix; });
}

// ---- search-free form: nearest-set-bit lookups in bitmaps of the adjacent-key prefix array ---------------------------------------
// For sorted UNIQUE keys clz(k_i ^ k_j) = min_{i <= m < j} delta_m, delta_m = clz(k_m ^ k_{m+1}): the radix tree is the Cartesian
// tree of the delta array.  Both searches of the reference (BVH.compute:39-47 and :74-89) return the LAST position at which a
// predicate that is monotone along the sorted keys still holds, hence (exactly, whatever probe sequence finds it)
//     dmin            = min(delta_{i-1}, delta_i)                         (delta_{-1} = delta_{n-1} = -1, BVH.compute:26-27)
//     the other end j = the nearest m in direction d with delta_m <= dmin (j = m going right, m + 1 going left)
//     split           = the first m >= first with delta_m <= clz(k_first ^ k_last)
// The workgroup forms B[v] = { m in its key window : delta_m <= v }, v = 0 .. 31, once: lane p of a 32-lane half holds the row
// 0xFFFFFFFF << delta_p (bit v set iff delta_p <= v) and a 32 x 32 bit transpose across the lanes (five ds_swizzle exchanges)
// leaves lane v holding 32 positions of B[v] — ~20 instructions per 64 positions.  A node is then two lookups (word pair, shift,
// count trailing / leading zeros) instead of two dependent probe loops of 9 (mean) .. 26 (a wave's widest node) LDS round trips at
// 45 % lane utilisation (25 us each at 1 M nodes, HISTORY.md §4).  A lookup that finds nothing inside the window hands the node to
// wide_node_search above, as a probe leaving the window did before.  tests/test_tree_bitmaps.py restates all of this on the CPU
// against the oracle's literal searches.
constexpr int kTreeWords = kTreeWindow / 32;          // 32-position words per B[v]
constexpr int kBitStride = 33;                        // words of one position block for v = 0 .. 31 (+1: conflict-free columns)
constexpr int kBitWords = (kTreeWords + 2) * kBitStride;      // one all-zero block in front and one behind: lookups read word pairs
constexpr int kBitSums = 34;                          // per v: one bit per non-empty word; [32]: "equal neighbours seen" (see below)

template <int J>
__device__ __forceinline__ uint32_t swizzle_xor(uint32_t x)        // lane l <- lane l ^ J inside each half of the wave
{
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, (J << 10) | 0x1F);
}

// one exchange step of the transpose: lanes l and l ^ J swap the half-blocks that belong to each other.  `keep` = the bits a lane
// keeps (M in the lower lane of a pair, ~M in the upper), `amt` = the right-rotation that moves the partner's bits into place
// (32 - J: a left shift by J for the lower lane, J for the upper — under ~keep the rotation IS the shift): exchange, v_alignbit,
// v_bfi — three instructions per step (the kernel is bound by its instruction count: ~1 200 per wave)
template <int S>
__device__ __forceinline__ uint32_t transpose_step(uint32_t x, uint32_t keep, uint32_t amt)
{
    constexpr uint32_t J = 1u << S;
    // lane ^ 1 and lane ^ 2 are quad permutes (DPP: the vector pipe); 4, 8, 16 go through ds_swizzle
    const uint32_t y = S == 0   ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true)      // quad_perm:[1,0,3,2]
                       : S == 1 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true)      // quad_perm:[2,3,0,1]
                                : swizzle_xor<(int)J>(x);
    const uint32_t moved = __builtin_amdgcn_alignbit(y, y, amt);
    return (x & keep) | (moved & ~keep);
}

struct transpose_consts { uint32_t keep[5], amt[5]; };
__device__ __forceinline__ transpose_consts make_transpose_consts(uint32_t lane)
{
    transpose_consts c;
    const uint32_t M[5] = {0x55555555u, 0x33333333u, 0x0F0F0F0Fu, 0x00FF00FFu, 0x0000FFFFu};
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const bool upper = (lane >> s) & 1u;
        c.keep[s] = upper ? ~M[s] : M[s];
        c.amt[s] = upper ? (1u << s) : 32u - (1u << s);
    }
    return c;
}

// lane v (of each 32-lane half) ends up with bit p = bit v of the row lane p came with
__device__ __forceinline__ uint32_t transpose32_lanes(uint32_t x, const transpose_consts& c)
{
    x = transpose_step<0>(x, c.keep[0], c.amt[0]);
    x = transpose_step<1>(x, c.keep[1], c.amt[1]);
    x = transpose_step<2>(x, c.keep[2], c.amt[2]);
    x = transpose_step<3>(x, c.keep[3], c.amt[3]);
    x = transpose_step<4>(x, c.keep[4], c.amt[4]);
    return x;
}

typedef __attribute__((address_space(3))) uint32_t lds_u32_rw;
struct delta_bitmaps {
    lds_u32* bits;                                    // [kBitWords]: word q of B[v] at (q + 1) * kBitStride + v
    lds_u32* sums;                                    // [kBitSums]
    __device__ __forceinline__ uint32_t word(int q, int v) const { return bits[(q + 1) * kBitStride + v]; }
    // window position of the first member of B[v] at or after position s (0 <= s < kTreeWindow)
    __device__ __forceinline__ bool first_from(int v, int s, int& pos) const
    {
        const int q = s >> 5;
        const u64 w = (((u64)word(q + 1, v) << 32) | word(q, v)) >> (s & 31);
        if (w) { pos = s + __builtin_ctzll(w); return true; }
        const uint32_t rest = sums[v] & ~((4u << q) - 1u);              // words past q + 1
        if (!rest) return false;
        const int c = __builtin_ctz(rest);
        pos = c * 32 + __builtin_ctz(word(c, v));
        return true;
    }
    // ... of the last member at or before position e (e may be -1)
    __device__ __forceinline__ bool last_upto(int v, int e, int& pos) const
    {
        if (e < 0) return false;
        const int q = e >> 5;
        const u64 w = (((u64)word(q, v) << 32) | word(q - 1, v)) << (31 - (e & 31));
        if (w) { pos = e - __builtin_clzll(w); return true; }
        const uint32_t rest = q >= 1 ? sums[v] & ((1u << (q - 1)) - 1u) : 0u;    // words before q - 1
        if (!rest) return false;
        const int c = 31 - __builtin_clz(rest);
        pos = c * 32 + 31 - __builtin_clz(word(c, v));
        return true;
    }
};

// before the barrier that publishes the staged keys: the two all-zero blocks and the summary words
__device__ __forceinline__ void clear_delta_bitmaps(uint32_t* s_bits, uint32_t* s_sums)
{
    const uint32_t t = threadIdx.x;
    if (t < (uint32_t)kBitStride) s_bits[t] = 0u;
    else if (t < 2u * kBitStride) s_bits[(kTreeWords + 1) * kBitStride + (t - kBitStride)] = 0u;
    else if (t < 2u * kBitStride + kBitSums) s_sums[t - 2u * kBitStride] = 0u;
}

// Stages the window's keys in LDS and forms the bitmaps on the way (one barrier must follow).  Wave w takes the 64-position
// chunks w, w + 4, w + 8: lane p's delta needs the key of position p + 1 — lane p + 1's (DPP wave_shl:1), the last lane loads it.
__device__ __forceinline__ void stage_keys_and_bitmaps(const uint32_t* __restrict__ codes, uint32_t* s_keys, uint32_t* s_bits,
                                                       uint32_t* s_sums, const key_window& win)
{
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    constexpr int kChunks = kTreeWindow / 64, kWaves = kTreeThreads / 64, kPer = kChunks / kWaves;
    static_assert(kChunks % kWaves == 0, "every wave takes the same number of chunks");
    uint32_t key[kPer], next[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int m = win.w0 + ((int)wave + k * kWaves) * 64 + (int)lane;
        key[k] = m < win.w1 ? codes[m] : 0u;
        next[k] = 0u;
        if (lane == 63u && m + 1 < win.w1) next[k] = codes[m + 1];
    }
    const transpose_consts tc = make_transpose_consts(lane);
    uint32_t some = 0u;                       // this lane's summary bits: lane (v, half) over the wave's chunks
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const int c = (int)wave + k * kWaves;
        const int p = c * 64 + (int)lane, m = win.w0 + p;
        if (m < win.w1) s_keys[p] = key[k];
        const uint32_t from_next_lane = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key[k], 0x130, 0xf, 0xf, false);    // wave_shl:1
        const uint32_t nxt = lane == 63u ? next[k] : from_next_lane;
        uint32_t row = 0u;                    // no such position, or delta unknown (the key behind the window): member of no B[v]
        if (m + 1 < win.w1) {
            const uint32_t x = key[k] ^ nxt;
            row = x ? 0xFFFFFFFFu << __builtin_clz(x) : 0u;
            if (!x) s_sums[32] = 1u;          // equal neighbours (a caller's raw keys): this window takes the probe loops
        } else if (m + 1 == win.num && m < win.w1) {
            row = 0xFFFFFFFFu;                // delta_{n-1} = -1 (BVH.compute:26-27): below every value
        }
        const uint32_t x = transpose32_lanes(row, tc);
        const int wd = 2 * c + (int)(lane >> 5);
        s_bits[(wd + 1) * kBitStride + (int)(lane & 31u)] = x;
        some |= x ? 1u << wd : 0u;
    }
    if (some) atomicOr(&s_sums[lane & 31u], some);
}

// ---- range hierarchy ------------------------------------------------------------------------------------------
// The reference refits bottom-up: one thread per leaf climbs to the root, the second arrival at a node merges the
// child boxes (BVH.compute:152-220).  The result is the union of the leaf AABBs in the node's leaf range, and min /
// max are exact, so ANY evaluation order gives the same floats.  A node of a Karras tree covers a CONTIGUOUS range
// [first, last] of the sorted leaves, and TreeConstructor has that range in its hands.  So inside lbvh_build_scene
// the refit is no climb at all: gather_hier_kernel writes the leaf boxes in sorted order plus the union of every
// aligned group of 2, 4, 8, ... leaves (a binary hierarchy, 2 x 32 B per leaf), and the tree kernel answers "box of
// [a, b]" with <= 2 boxes per level (the classic bottom-up segment-tree walk): no arrival counters, no dependent
// chain over tree levels, no second kernel, every node independent.  Both trees (the reference's and the derived
// traversal tree) are over the same sorted leaves, so they share one hierarchy.
// Layout: level k (blocks of 2^k leaves) starts at box index 2 N2 - (2 N2 >> k), N2 = the leaf count rounded up to a
// power of two; only blocks that start below n are ever written or read.  Minima and maxima live in two arrays of
// packed 12-byte corners: the queries are gathers, and what they cost is the number of distinct cache lines a wave's
// load touches — 5 corners per 64-byte line instead of 2 boxes.
typedef float f3 __attribute__((ext_vector_type(3)));
struct corner3 { float x, y, z; };                  // 12 bytes, 4-byte aligned
struct hier_t {
    corner3* lo;            // [2 N2] minima
    corner3* hi;            // [2 N2] maxima
    uint32_t n2x2;          // 2 * N2
    uint32_t levels;        // log2(N2): the top level has one block
};

__device__ __forceinline__ uint32_t hier_offset(uint32_t n2x2, uint32_t k) { return n2x2 - (n2x2 >> k); }

__device__ __forceinline__ f3 load_corner(const corner3* p) { return *reinterpret_cast<const f3 __attribute__((aligned(4)))*>(p); }
__device__ __forceinline__ void store_corner(corner3* p, float x, float y, float z)
{
    f3 v = {x, y, z};
    *reinterpret_cast<f3 __attribute__((aligned(4)))*>(p) = v;
}

// Unions of the leaf boxes a[q] .. b[q] (inclusive) for NQ ranges at once.  The walk over the levels needs no loaded
// data (block indices come from a and b alone), so LV levels are resolved per round: all their block indices first,
// then all loads back to back (a block that is not part of the range is marked by the index of the NEUTRAL box, 2 N2 - 1, and
// its load is masked off), then the min / max — one memory latency per LV levels instead of one per
// block (the straightforward loop measured 17 us per query at 1 M nodes, all of it waiting).
__device__ __forceinline__ void wave_union(float mn[3], float mx[3])
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], d));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d));
        }
    }
}

typedef uint32_t u3v __attribute__((ext_vector_type(3)));
// BUF: the corners are fetched through buffer descriptors with 32-bit byte offsets (v_mul_u32_u24 + buffer_load_dwordx3) instead
// of 64-bit flat addresses (one v_mad_u64_u32 per load: the kernel is bound by its instruction count, and that one is a slow
// instruction) — whenever the hierarchy's 12 n2x2 bytes fit 32 bits (up to 178 M leaves)
template <int NQ, int LV, bool BUF>
__device__ __forceinline__ void range_boxes_impl(const hier_t& h, const uint32_t a[NQ], const uint32_t b[NQ], float mn[NQ][3],
                                                 float mx[NQ][3])
{
    const uint32_t neutral = h.n2x2 - 1u;
    __amdgpu_buffer_rsrc_t lo_rsrc, hi_rsrc;
    if (BUF) {
        lo_rsrc = __builtin_amdgcn_make_buffer_rsrc(h.lo, 0, (int)(h.n2x2 * 12u), 0x00020000);
        hi_rsrc = __builtin_amdgcn_make_buffer_rsrc(h.hi, 0, (int)(h.n2x2 * 12u), 0x00020000);
    }
    uint32_t l[NQ], r[NQ];
    bool more = false;
    // the corners of a round's blocks.  Set to the neutral element ONCE: a load is masked off where a block is not part of the
    // range, and what such a register then still holds is a block of an EARLIER round of the same range — taking it again changes
    // nothing (min / max are idempotent), so the rounds need no re-initialisation (36 v_mov per round of the reference's query)
    f3 lo[NQ][LV][2], hi[NQ][LV][2];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        l[q] = a[q];
        r[q] = b[q] + 1u;
        more = more || l[q] < r[q];
#pragma unroll
        for (int d = 0; d < 3; d++) { mn[q][d] = INFINITY; mx[q][d] = -INFINITY; }
#pragma unroll
        for (int v = 0; v < LV; v++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const f3 pinf = {INFINITY, INFINITY, INFINITY}, ninf = {-INFINITY, -INFINITY, -INFINITY};
                lo[q][v][e] = pinf;
                hi[q][v][e] = ninf;
            }
    }
    for (uint32_t k = 0; more; k += LV) {
        uint32_t idx[NQ][LV][2];
        more = false;
#pragma unroll
        for (int v = 0; v < LV; v++) {
            const uint32_t lev = min(k + (uint32_t)v, 31u);
            const uint32_t off = hier_offset(h.n2x2, lev);
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const bool tl = l[q] < r[q] && (l[q] & 1u);
                idx[q][v][0] = tl ? off + l[q] : neutral;
                l[q] += tl ? 1u : 0u;
                const bool tr = l[q] < r[q] && (r[q] & 1u);
                r[q] -= tr ? 1u : 0u;
                idx[q][v][1] = tr ? off + r[q] : neutral;
                l[q] >>= 1;
                r[q] >>= 1;
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int v = 0; v < LV; v++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    // a block that is not part of the range: no load at all (exec-masked).  Round 2 read a NEUTRAL box there instead
                    // of branching; but what a gather costs is the data its ACTIVE lanes return (768 B per full dwordx3 wave load
                    // through a 64 B / clk path), and two thirds of these loads had nothing to fetch: 78.5 -> 74.3 us at 1 M nodes
                    if (idx[q][v][e] != neutral) {
                        if (BUF) {
                            const u3v x = __builtin_amdgcn_raw_buffer_load_b96(lo_rsrc, idx[q][v][e] * 12u, 0, 0);
                            const u3v y = __builtin_amdgcn_raw_buffer_load_b96(hi_rsrc, idx[q][v][e] * 12u, 0, 0);
                            lo[q][v][e].x = __uint_as_float(x.x); lo[q][v][e].y = __uint_as_float(x.y); lo[q][v][e].z = __uint_as_float(x.z);
                            hi[q][v][e].x = __uint_as_float(y.x); hi[q][v][e].y = __uint_as_float(y.y); hi[q][v][e].z = __uint_as_float(y.z);
                        } else {
                            lo[q][v][e] = load_corner(h.lo + idx[q][v][e]);
                            hi[q][v][e] = load_corner(h.hi + idx[q][v][e]);
                        }
                    }
                }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
#pragma unroll
            for (int v = 0; v < LV; v++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    mn[q][0] = fminf(mn[q][0], lo[q][v][e].x); mn[q][1] = fminf(mn[q][1], lo[q][v][e].y);      // MergeAABB :152-170
                    mn[q][2] = fminf(mn[q][2], lo[q][v][e].z);
                    mx[q][0] = fmaxf(mx[q][0], hi[q][v][e].x); mx[q][1] = fmaxf(mx[q][1], hi[q][v][e].y);
                    mx[q][2] = fmaxf(mx[q][2], hi[q][v][e].z);
                }
            more = more || l[q] < r[q];
        }
    }
}

template <int NQ, int LV>
__device__ __forceinline__ void range_boxes(const hier_t& h, const uint32_t a[NQ], const uint32_t b[NQ], float mn[NQ][3],
                                            float mx[NQ][3])
{
    if (h.n2x2 <= 0x15555555u) range_boxes_impl<NQ, LV, true>(h, a, b, mn, mx);        // 12 n2x2 < 2^32
    else range_boxes_impl<NQ, LV, false>(h, a, b, mn, mx);
}

// hierarchy levels resolved per round of a range query: one range (the reference's boxes) / two ranges (the derived tree's child
// boxes).  Round 5: 3 / 1 instead of round 2's 4 / 2 — fewer block indices and corners alive at once: 72 -> 60 - 62 registers, 7 -> 8
// waves per SIMD, tree_pair_kernel 86.4 -> 81.8 us at 1 M triangles (4 / 1, 2 / 1, 1 / 1 within 0.5 us; 4 / 3: 89, 4 / 4: 94),
// the 16 M-triangle build unchanged.  Round 6: the derived tree's two ranges are queried one after the other (tree_body), each
// LBVH_RQ_LV2 = 3 levels per round: 73.6 -> 72.0 us (2: 74.2, 4: 81.7, 5: 93.8)
#ifndef LBVH_RQ_LV1
#define LBVH_RQ_LV1 3
#endif
#ifndef LBVH_RQ_LV2
#define LBVH_RQ_LV2 3
#endif
__device__ __noinline__ void codes_sink(lbvh_internal_node* a, lbvh_fast_node* b)
{
    if (a) reinterpret_cast<uint32_t*>(a)[5] = 1u;
    if (b) reinterpret_cast<uint32_t*>(b)[0] = 1u;
}
#ifdef LBVH_TREE_CLOCK
__device__ unsigned long long g_tree_clock[2][16];       // [tree mode - 1][phase]: wave-cycles summed over all waves
#define TREE_TICK(ph) do { const long long now_ = clock64(); if (lane_id() == 0 && MODE != TREE_TOPOLOGY) atomicAdd(&g_tree_clock[MODE - 1][ph], (unsigned long long)(now_ - tick_)); tick_ = now_; } while (0)
#else
#define TREE_TICK(ph) do { } while (0)
#endif
enum { TREE_TOPOLOGY = 0, TREE_REFERENCE = 1, TREE_FUSED = 2 };
// measurement builds only (tools/build_variant.sh tree_eN -DLBVH_TREE_EXP=N): 1 = no node / leaf stores, 2 = no range queries,
// 4 = no box / traversal-node output; LBVH_TREE_LOOPS: the probe loops instead of the bitmap lookups
#ifndef LBVH_TREE_EXP
#define LBVH_TREE_EXP 0
#endif
#ifdef LBVH_TREE_LOOPS
constexpr bool kTreeLookups = false;
#else
constexpr bool kTreeLookups = true;
#endif
// TREE_TOPOLOGY   lbvh_build_tree: the reference's node arrays only
// TREE_REFERENCE  + bvh[i] = box of the node's range (BVHData, BVH.compute:215)
// TREE_FUSED      the derived traversal tree: nothing but the 64-byte traversal node (both child boxes + child references)
// s_keys[kTreeWindow], s_out[kTreeThreads * kQuads of the mode]: the calling kernel's LDS (tree_pair_kernel runs two modes
// in one launch on ONE set)
template <int MODE, bool LOOKUPS = kTreeLookups>
__device__ __forceinline__ void tree_body(uint32_t block, uint32_t* s_keys, uint32_t* s_bits, uint32_t* s_sums, float4* s_out,
                                          const uint32_t* __restrict__ codes,
                                          uint32_t n, lbvh_internal_node* __restrict__ internal, lbvh_leaf_node* __restrict__ leaf,
                                          uint32_t* __restrict__ zero_word, hier_t hier, lbvh_aabb* __restrict__ bvh,
                                          lbvh_fast_node* __restrict__ fused, uint32_t leaf_base,
                                          const uint32_t* __restrict__ sorted_indices)
{
#ifdef LBVH_TREE_CLOCK
    long long tick_ = clock64();
#endif
    const uint32_t thread_id = block * kTreeThreads + threadIdx.x;
    if (thread_id == 0 && zero_word) *zero_word = 0u;      // lbvh_build_tree + lbvh_refit: the refit's frontier counter
    key_window win;
    win.lds = (lds_u32*)s_keys;
    win.num = (int)n;
    win.w0 = max((int)(block * kTreeThreads) - kTreeHalo, 0);
    win.w1 = min((int)(block * kTreeThreads) + kTreeThreads + kTreeHalo, (int)n);
    if (LOOKUPS) {
        clear_delta_bitmaps(s_bits, s_sums);
        __syncthreads();                                   // (nothing is in flight yet: the waves are here within cycles of each other)
        stage_keys_and_bitmaps(codes, s_keys, s_bits, s_sums, win);
    } else {
        for (int k = win.w0 + (int)threadIdx.x; k < win.w1; k += kTreeThreads) s_keys[k - win.w0] = codes[k];
    }
    __syncthreads();
    TREE_TICK(0);
    // (no early return: the boxes leave through a workgroup-wide LDS transpose below)
    const bool in_range = thread_id < n - 1;                                               // :101
    const int idx = in_range ? (int)thread_id : win.w0;    // threads past the last node: a key INSIDE the window (results unused)
    const uint32_t self = win.at(idx);
    bool wide = false;             // a probe left the window: the node is searched by the whole wave further down

    // DetermineRange :35-52
    const int dl = win.delta(self, idx - 1, wide);
    const int dr = win.delta(self, idx + 1, wide);
    const int diff = dr - dl;
    const int d = (diff > 0) - (diff < 0);                                                 // sign(), :37
    const int dmin = d > 0 ? dl : (d < 0 ? dr : clz32(0u));                                // :38
    int first, last, split;
    // (equal neighbours somewhere in the window — possible only for a caller's raw keys, lbvh_build_tree — or d == 0: the loops)
    const bool lookups = LOOKUPS && s_sums[32] == 0u;
    if (LBVH_TREE_EXP & 32) {
        first = idx; last = idx + (int)(self & 3u); split = idx;
    } else if (lookups) {
        // "search-free form" above: the other end of the range and the split by nearest-set-bit lookups
        const delta_bitmaps bm = {(lds_u32*)s_bits, (lds_u32*)s_sums};
        const int p0 = idx - win.w0;
        int j = idx, pos = 0;
        if (d > 0) {
            if (dmin < 0) {                                      // the root: its range ends at delta_{n-1} = -1
                if (win.w1 == win.num) j = win.num - 1; else wide = true;
            } else if (bm.first_from(dmin, p0, pos)) {
                j = win.w0 + pos;
            } else {
                wide = true;
            }
        } else if (d < 0) {
            if (bm.last_upto(dmin, p0 - 1, pos)) j = win.w0 + pos + 1;
            else if (win.w0 == 0) j = 0;                         // delta_{-1} = -1
            else wide = true;
        }
        first = min(idx, j);                                                               // :50
        last = max(idx, j);
        split = first;
        if (!wide) {                                             // FindSplit :54-92 (first and last are inside the window)
            const uint32_t x = win.at(first) ^ win.at(last);
            if (x == 0u) split = (first + last) >> 1;                                      // :61-62
            else if (bm.first_from(__builtin_clz(x), first - win.w0, pos)) split = win.w0 + pos;
            else wide = true;                                    // (unreachable for sorted unique keys)
        }
    } else {
        uint32_t lmax = 2;                                                                 // :39
        while (in_range && win.delta(self, idx + (int)(lmax * (uint32_t)d), wide) > dmin) lmax *= 2;       // :40-41
        int l = 0;
        for (uint32_t t = lmax / 2; t >= 1 && in_range && !wide; t /= 2) {                 // :43
            if (win.delta(self, idx + (int)(((uint32_t)l + t) * (uint32_t)d), wide) > dmin)
                l += (int)t;                                                               // :45-46
        }
        const int j = idx + l * d;                                                         // :49
        first = min(idx, j);                                                               // :50
        last = max(idx, j);

        // FindSplit :54-92
        split = first;
        if (in_range && !wide) {
            if (!win.inside(first) || !win.inside(last)) {
                wide = true;
            } else {
                const uint32_t first_code = win.at(first);
                const uint32_t last_code = win.at(last);
                if (first_code == last_code) {
                    split = (first + last) >> 1;                                           // :61-62
                } else {
                    const int common_prefix = clz32(first_code ^ last_code);               // :67
                    split = first;
                    int step = last - first;
                    do {                                   // (probes stay inside [first, last], hence inside the window)
                        step = (step + 1) >> 1;                                            // :78
                        const int new_split = split + step;
                        if (new_split < last) {
                            const int split_prefix = clz32(first_code ^ win.at(new_split));
                            if (split_prefix > common_prefix) split = new_split;           // :85-86
                        }
                    } while (step > 1);
                }
            }
        }
    }
    TREE_TICK(1);
    // the wave's wide nodes, one after the other, every lane probing
    for (uint64_t todo = (LBVH_TREE_EXP & 8) ? 0ull : __ballot(in_range && wide); todo != 0; todo &= todo - 1) {
        const int src = __builtin_ctzll(todo);
        const int widx = __builtin_amdgcn_readlane(idx, src);
        const uint32_t wself = (uint32_t)__builtin_amdgcn_readlane((int)self, src);
        const int wd = __builtin_amdgcn_readlane(d, src), wdmin = __builtin_amdgcn_readlane(dmin, src);
        int f, la, sp;
        wide_node_range(codes, (int)n, widx, wself, wd, wdmin, f, la);
        const uint32_t other = codes[widx == f ? la : f];
        const uint32_t fc = widx == f ? wself : other, lc = widx == f ? other : wself;
        // the split from the window's bitmaps when the left child's range ends inside the window (it is the first member of
        // B[delta_node] at or after `first`, wherever the range ends)
        int pos = 0;
        if (lookups && fc != lc && win.inside(f) &&
            delta_bitmaps{(lds_u32*)s_bits, (lds_u32*)s_sums}.first_from(__builtin_clz(fc ^ lc), f - win.w0, pos) && win.w0 + pos < la)
            sp = win.w0 + pos;
        else
            sp = wide_node_split(codes, f, la, fc, lc);
        if ((int)lane_id() == src) { first = f; last = la; split = sp; }
    }
    TREE_TICK(2);
    const bool valid = in_range && !(split < 0 || (uint32_t)split + 1u >= n);   // invalid: only reachable with non-unique keys

    const bool left_leaf = split == first;                                                 // :114
    const bool right_leaf = split + 1 == last;                                             // :132
    if ((LBVH_TREE_EXP & 1) && valid && (first ^ last ^ split) == 0x7FFFFFF1) codes_sink(internal, fused);
    if (MODE != TREE_FUSED && valid && !(LBVH_TREE_EXP & 1)) {
        uint32_t* node = reinterpret_cast<uint32_t*>(&internal[thread_id]);
        // Inside lbvh_build_scene (TREE_REFERENCE) the reference's arrays are written as streaming data, here and below:
        // the frame that follows a rebuild walks the DERIVED scene, and 64 MB of node words and boxes written last would
        // push its lines out of the caches — trace part of a cfg2 step 0.245 -> 0.230 ms, the rebuild itself unchanged.
        // (lbvh_build_tree alone, TREE_TOPOLOGY, is followed by lbvh_refit, which reads the nodes: ordinary stores.)
        typedef uint32_t u2v __attribute__((ext_vector_type(2)));
        auto st2 = [](void* p, uint32_t x, uint32_t y) {
            const u2v v = {x, y};
            if (MODE == TREE_REFERENCE) __builtin_nontemporal_store(v, reinterpret_cast<u2v*>(p));
            else *reinterpret_cast<u2v*>(p) = v;
        };
        auto st1 = [](uint32_t* p, uint32_t x) {
            if (MODE == TREE_REFERENCE) __builtin_nontemporal_store(x, p);
            else *p = x;
        };
        // leftNode, leftNodeType, rightNode, rightNodeType as two 8-byte stores (node stride 24 B)
        st2(node + 0, (uint32_t)split, left_leaf ? LBVH_LEAF_NODE : LBVH_INTERNAL_NODE);
        st2(node + 2, (uint32_t)split + 1u, right_leaf ? LBVH_LEAF_NODE : LBVH_INTERNAL_NODE);
        st1(&node[5], thread_id);                                                          // index :111
        if (left_leaf) st2(&leaf[split], thread_id, (uint32_t)split);                      // :116-120
        else st1(&internal[split].parent, thread_id);                                      // :126
        if (right_leaf) st2(&leaf[split + 1], thread_id, (uint32_t)split + 1u);
        else st1(&internal[split + 1].parent, thread_id);                                  // :144
    }
    TREE_TICK(3);
    // The boxes: computed per node, written per LINE — every thread parks its record in LDS and its WAVE writes the wave's 64
    // records as consecutive float4 (a wave's store covers 1 KB of whole records, not 64 quarter lines).
    constexpr int kQuads = MODE == TREE_FUSED ? 4 : 2;             // float4 per record: 64-byte traversal node / 32-byte AABB
    if (MODE == TREE_REFERENCE) {
        const uint32_t a[1] = {(uint32_t)first}, b[1] = {valid ? (uint32_t)last : (uint32_t)first};
        float mn[1][3], mx[1][3];
        if (LBVH_TREE_EXP & 2) { for (int k = 0; k < 3; k++) { mn[0][k] = (float)first; mx[0][k] = (float)last; } }
        else range_boxes<1, LBVH_RQ_LV1>(hier, a, b, mn, mx);
        s_out[threadIdx.x * 2 + 0] = make_float4(mn[0][0], mn[0][1], mn[0][2], 0.0f);      // :215
        s_out[threadIdx.x * 2 + 1] = make_float4(mx[0][0], mx[0][1], mx[0][2], 0.0f);
    }
    if (MODE == TREE_FUSED) {
        const uint32_t a[2] = {(uint32_t)first, (uint32_t)split + 1u};
        const uint32_t b[2] = {(uint32_t)split, valid ? (uint32_t)last : (uint32_t)split};
        float cmn[2][3], cmx[2][3];
        if (LBVH_TREE_EXP & 2) { for (int k = 0; k < 3; k++) { cmn[0][k] = cmn[1][k] = (float)first; cmx[0][k] = cmx[1][k] = (float)last; } }
        else {
            // the two child ranges one after the other, LBVH_RQ_LV2 levels per round each (round 6: 3) — not both at once with one
            // level per round (rounds 2 - 5): a round is a full memory latency, the waves wait 64 % of their cycles
            // (profiles/r6/tree_pair_counters.txt), and two queries of three levels keep as many corners in registers as the
            // reference's one (four levels per round: 62 -> 70 registers, one wave per SIMD less, 72 -> 82 us)
            range_boxes<1, LBVH_RQ_LV2>(hier, &a[0], &b[0], &cmn[0], &cmx[0]);
            range_boxes<1, LBVH_RQ_LV2>(hier, &a[1], &b[1], &cmn[1], &cmx[1]);
        }
        // child reference: a line index — node index, or LEAF | leaf_base + the triangle's ORIGINAL index (the triangle
        // lines stay in the caller's order: lbvh_common.h)
        uint32_t lref = (uint32_t)split, rref = (uint32_t)split + 1u;
        if (valid && left_leaf) lref = 0x80000000u | (leaf_base + sorted_indices[split]);
        if (valid && right_leaf) rref = 0x80000000u | (leaf_base + sorted_indices[split + 1]);
        s_out[threadIdx.x * 4 + 0] = make_float4(cmn[0][0], cmn[0][1], cmn[0][2], __uint_as_float(lref));
        s_out[threadIdx.x * 4 + 1] = make_float4(cmx[0][0], cmx[0][1], cmx[0][2], __uint_as_float(rref));
        // (the two spare words: the children's leaf positions, read by LBVH_TRACE_FAST_EXACT when the child is a leaf)
        s_out[threadIdx.x * 4 + 2] = make_float4(cmn[1][0], cmn[1][1], cmn[1][2], __uint_as_float((uint32_t)split));
        s_out[threadIdx.x * 4 + 3] = make_float4(cmx[1][0], cmx[1][1], cmx[1][2], __uint_as_float((uint32_t)split + 1u));
    }
    TREE_TICK(4);
    if (MODE != TREE_TOPOLOGY) {
        // a wave's 64 records are one contiguous piece of the output and of the staging area: the wave writes them out itself, without
        // waiting for the workgroup's other waves (LDS executes a wave's instructions in order; round 6: the workgroup-wide barrier
        // that stood here made every wave wait for the one searching a wide node in memory — 73.3 -> 72.2 us)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        TREE_TICK(5);
        const uint32_t b0 = block * kTreeThreads;
        float4* out = MODE == TREE_FUSED ? reinterpret_cast<float4*>(fused + b0) : reinterpret_cast<float4*>(bvh + b0);
        const uint32_t live = n - 1 > b0 ? min(n - 1 - b0, (uint32_t)kTreeThreads) * kQuads : 0u;   // float4s of existing nodes
        const uint32_t wbase = (threadIdx.x >> 6) * 64u * kQuads;
#pragma unroll
        for (int k = 0; k < kQuads; k++) {
            const uint32_t q = wbase + (uint32_t)k * 64u + lane_id();
            if ((LBVH_TREE_EXP & 4) && s_out[q].x != 12345.678f) continue;
            if (q < live) {
                if (MODE == TREE_REFERENCE) lbvh_nt_store(&out[q], s_out[q]);
                else out[q] = s_out[q];
            }
        }
    }
    TREE_TICK(6);
}

template <int MODE>
__global__ __launch_bounds__(kTreeThreads) void tree_kernel(const uint32_t* __restrict__ codes, uint32_t n,
                                                            lbvh_internal_node* __restrict__ internal,
                                                            lbvh_leaf_node* __restrict__ leaf, uint32_t* __restrict__ zero_word,
                                                            hier_t hier, lbvh_aabb* __restrict__ bvh,
                                                            lbvh_fast_node* __restrict__ fused, uint32_t leaf_base,
                                                            const uint32_t* __restrict__ sorted_indices)
{
    __shared__ uint32_t s_keys[kTreeWindow];
    __shared__ float4 s_out[MODE == TREE_TOPOLOGY ? 1 : kTreeThreads * (MODE == TREE_FUSED ? 4 : 2)];
    __shared__ uint32_t s_bits[kBitWords];
    __shared__ uint32_t s_sums[kBitSums];
    tree_body<MODE>(blockIdx.x, s_keys, s_bits, s_sums, s_out, codes, n, internal, leaf, zero_word, hier, bvh, fused, leaf_base, sorted_indices);
}

// lbvh_build_scene's two trees — the derived one over aligned keys, the reference's over the distributed keys — in ONE launch,
// workgroups alternating (even: derived, odd: reference): both progress side by side as they did on two streams, without the
// cross-stream dependencies (fork, hierarchy-ready, join: ~5 us each on this runtime) around them
__global__ __launch_bounds__(kTreeThreads) void tree_pair_kernel(const uint32_t* __restrict__ aligned_codes,
                                                                 const uint32_t* __restrict__ codes, uint32_t n,
                                                                 lbvh_internal_node* __restrict__ internal,
                                                                 lbvh_leaf_node* __restrict__ leaf, hier_t hier,
                                                                 lbvh_aabb* __restrict__ bvh, lbvh_fast_node* __restrict__ fused,
                                                                 uint32_t leaf_base, const uint32_t* __restrict__ sorted_indices)
{
    __shared__ uint32_t s_keys[kTreeWindow];
    __shared__ float4 s_out[kTreeThreads * 4];
    __shared__ uint32_t s_bits[kBitWords];
    __shared__ uint32_t s_sums[kBitSums];
    const uint32_t block = blockIdx.x >> 1;
    if (blockIdx.x & 1u)
        tree_body<TREE_REFERENCE>(block, s_keys, s_bits, s_sums, s_out, codes, n, internal, leaf, nullptr, hier, bvh, nullptr, 0u, nullptr);
    else
        tree_body<TREE_FUSED>(block, s_keys, s_bits, s_sums, s_out, aligned_codes, n, nullptr, nullptr, nullptr, hier, nullptr, fused, leaf_base,
                              sorted_indices);
}

// ---------------------------------------------------------------------------------------------
// a-8  BVHConstructor (bottom-up refit)       BVH.compute:152-220
// The reference: one thread per leaf walks to the root; the second thread to arrive at a node (per-node
// counter, InterlockedCompareExchange) merges the child boxes — with no fence between a thread's box
// store and the sibling's read (BVH.compute:185-215).  The result is the union of the leaf AABBs under each
// node, and min/max are exact, so ANY evaluation order gives the same floats.  On MI355X a CU's L1 is never
// refreshed by other CUs' stores and the 8 XCD L2s are not coherent with each other, so a cross-workgroup
// hand-off per tree level costs a write-through store, a device-scope atomic and L1-bypassing loads — about
// three trips to the coherence point per level, ~12 levels deep above a 1024-leaf subtree: that chain measured
// 81 of the kernel's 128 us at 1 M leaves (fence pair per level: 3.99 ms; all-global sc1 walk: 0.18 ms).
// Here NO hand-off crosses a workgroup:
//   refit_kernel    one workgroup per 1024 consecutive leaves; a thread carries its subtree's box and leaf
//                   range upward, merging through LDS arrival counters and LDS-parked sibling boxes while
//                   the parent's subtree stays inside the workgroup's leaves (> 95 % of all nodes).  It also
//                   emits the union box of every aligned group of 64 and of 1024 leaves (range levels 1, 2).
//   refit_levels    one workgroup: range levels 3.. (aligned groups of 16^k * 64 leaves).
//   refit_ranges    one thread per internal node recovers the node's leaf range from the node arrays alone
//                   (see below) and appends the FRONTIER nodes — range crossing a 1024-leaf boundary, i.e.
//                   exactly the nodes refit_kernel could not finish — to a list.
//   refit_frontier  one wave per frontier node, all independent: its box is the union of <= 63 + 63 leaves
//                   and <= 15 + 15 entries per range level covering [first, last] — no dependence on any
//                   other frontier node, so no chain, no atomics, no sc1 traffic.
// Leaf range of node p from the arrays (Karras numbering): p is one end of its range and its split is its
// left child's index g; p <= g means p is the FIRST leaf (p is the root or a right child), else the LAST
// (a left child).  A right child shares its parent's last leaf, a left child its parent's first; so the
// unknown end is the index of the nearest ancestor of the OTHER kind (or 0 / n-1 above the root): a walk
// over consecutive same-kind ancestors, 2 nodes on average, bounded by the tree depth.

__device__ __forceinline__ void load_box_plain(const lbvh_aabb* p, float mn[3], float mx[3])
{
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1];
    mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
    mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
}

__device__ __forceinline__ void store_box_plain(lbvh_aabb* p, const float mn[3], const float mx[3])
{
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = make_float4(mn[0], mn[1], mn[2], 0.0f);                                         // :215
    q[1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
}

constexpr int kRefitThreads = 1024;     // leaves per workgroup = range level 2
constexpr int kLevel1Shift = 6;         // range level 1 = 64 leaves (one wave of refit_kernel)
constexpr int kFanShift = 4;            // level k + 1 = 16 entries of level k
constexpr int kMaxLevels = 8;           // 64 * 16^7 > 2^31 leaves

struct refit_levels_t {
    lbvh_aabb* box[kMaxLevels + 1];     // [1..levels]; box[k][e] = union of leaves [e * g_k, (e + 1) * g_k) below n
    uint32_t count[kMaxLevels + 1];
    int levels;
};

// The stand-alone lbvh_refit (a caller's tree, no keys at hand): the climb described above.
// sorted_indices == nullptr: tri_aabb is already in sorted (leaf) order.
__global__ __launch_bounds__(kRefitThreads) void refit_kernel(uint32_t n, const lbvh_internal_node* __restrict__ internal,
                                                              const lbvh_leaf_node* __restrict__ leaf,
                                                              const lbvh_aabb* __restrict__ tri_aabb,
                                                              const uint32_t* __restrict__ sorted_indices,
                                                              lbvh_aabb* bvh, refit_levels_t lv,
                                                              uint32_t* __restrict__ frontier_count,
                                                              uint32_t* __restrict__ frontier_list)
{
    // parked child boxes, [side][slot] = {min xyz, range word | max xyz, -}: two 16-byte LDS accesses per box.
    // range word: first | last << 16 relative to the workgroup, bit 31 = the parked child is a leaf.   64 KB
    __shared__ float4 s_box[2][kRefitThreads][2];
    __shared__ uint32_t s_flag[kRefitThreads];          // +1 = left child arrived, +0x10000 = right child
    __shared__ float s_wave[6][kRefitThreads / LBVH_WAVE];
    __shared__ uint2 s_node[kRefitThreads];             // {leftNode (= split), parent} of this index block's nodes: 8 KB
    __shared__ uint32_t s_front_n, s_front_base;
    const uint32_t t = threadIdx.x;
    const uint32_t b0 = blockIdx.x * (uint32_t)kRefitThreads;
    const uint32_t j = b0 + t;
    s_flag[t] = 0;
    if (t == 0) s_front_n = 0;
    // the climb below only ever reads nodes of this index block: stage their two words once, coalesced, so that
    // the dependent chain (one node per level) runs on LDS latency instead of one L2 / HBM round trip per level
    if (j < n - 1) {
        const uint32_t* nd = reinterpret_cast<const uint32_t*>(&internal[j]);
        s_node[t] = make_uint2(nd[0], nd[4]);
    }
    __syncthreads();

    uint32_t q = 0xFFFFFFFFu, child_id = j, first = j, last = j;
    uint32_t child_leaf = 0x80000000u;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (j < n) {                                                                           // :179
        load_box_plain(&tri_aabb[sorted_indices ? sorted_indices[j] : j], mn, mx);
        q = leaf[j].parent;                                                                // :181
    }
    {   // range levels 1 and 2 from the leaf boxes (slots past n are neutral)
        float wmn[3] = {mn[0], mn[1], mn[2]}, wmx[3] = {mx[0], mx[1], mx[2]};
        wave_union(wmn, wmx);
        if ((t & 63u) == 0) {
            if ((j >> kLevel1Shift) < lv.count[1]) store_box_plain(&lv.box[1][j >> kLevel1Shift], wmn, wmx);
#pragma unroll
            for (int k = 0; k < 3; k++) { s_wave[k][t >> 6] = wmn[k]; s_wave[3 + k][t >> 6] = wmx[k]; }
        }
        __syncthreads();
        if (t < 64u) {
            float bmn[3], bmx[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                bmn[k] = t < (uint32_t)(kRefitThreads / LBVH_WAVE) ? s_wave[k][t] : INFINITY;
                bmx[k] = t < (uint32_t)(kRefitThreads / LBVH_WAVE) ? s_wave[3 + k][t] : -INFINITY;
            }
            wave_union(bmn, bmx);
            if (t == 0 && lv.levels >= 2) store_box_plain(&lv.box[2][blockIdx.x], bmn, bmx);
        }
    }
    if (j < n) {
        for (int guard = 0; q != 0xFFFFFFFFu && guard < 64; guard++) {
            if (q >= n - 1) break;
            // LOCAL arrival: q's index and the carried range lie inside the workgroup's 1024 indices.  Anything
            // else is a frontier node (refit_frontier computes it from the range levels).
            const bool local = q >= b0 && q - b0 < (uint32_t)kRefitThreads && first >= b0 &&
                               last - b0 < (uint32_t)kRefitThreads;
            if (!local) break;
            const uint32_t slot = q - b0;
            const uint2 nd = s_node[slot];
            const uint32_t next = nd.y;
            // Karras numbering: the children of q are indices split and split + 1, so a child (leaf or internal)
            // is the left one iff its index is the split
            const uint32_t side = nd.x == child_id ? 0u : 1u;
            // park my box and range, THEN arrive (LDS executes in issue order: the sibling that draws
            // the second ticket finds them)
            s_box[side][slot][0] = make_float4(mn[0], mn[1], mn[2],
                                               __uint_as_float((first - b0) | ((last - b0) << 16) | child_leaf));
            s_box[side][slot][1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
            const uint32_t old = atomicAdd(&s_flag[slot], side == 0 ? 1u : 0x10000u);      // :185 (LDS)
            if (old == 0) break;                                                           // first arrival :186-189
            // second arrival: merge and carry on.  Nothing is stored here: both children's boxes stay parked in this
            // slot, and the node is written from them, coalesced by node index, after the climb.
            const uint32_t o = side ^ 1u;
            const float4 omn = s_box[o][slot][0], omx = s_box[o][slot][1];
            const uint32_t orange = __float_as_uint(omn.w);
            mn[0] = fminf(mn[0], omn.x); mn[1] = fminf(mn[1], omn.y); mn[2] = fminf(mn[2], omn.z);   // MergeAABB :152-170
            mx[0] = fmaxf(mx[0], omx.x); mx[1] = fmaxf(mx[1], omx.y); mx[2] = fmaxf(mx[2], omx.z);
            first = min(first, b0 + (orange & 0xFFFFu));
            last = max(last, b0 + ((orange >> 16) & 0x7FFFu));
            if (q == 0) break;                          // the root: its parent word is whatever the caller's buffer held
            child_id = q;
            child_leaf = 0u;
            q = next;                                                                      // :217
        }
    }
    // internal nodes of this index block that did not see both children arrive through LDS are the frontier
    // (one global atomic per workgroup: the list order is irrelevant)
    __syncthreads();
    const bool is_front = j < n - 1 && s_flag[t] != 0x10001u;
    uint32_t my = 0;
    if (is_front) my = atomicAdd(&s_front_n, 1u);
    // finished nodes (both children arrived): their children's boxes are the two parked entries of their slot.  Written
    // with 2 threads per node — thread t takes float4 (t & 1) of node (t >> 1) + 512 k — so that a wave's store covers whole
    // consecutive records instead of 64 half lines
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) {
        const uint32_t slot = (t >> 1) + 512u * k, h = t & 1u, node = b0 + slot;
        if (node >= n - 1 || s_flag[slot] != 0x10001u) continue;
        const float4 l = s_box[0][slot][h], r = s_box[1][slot][h];
        reinterpret_cast<float4*>(&bvh[node])[h] =                                     // :215
            h == 0u ? make_float4(fminf(l.x, r.x), fminf(l.y, r.y), fminf(l.z, r.z), 0.0f)
                    : make_float4(fmaxf(l.x, r.x), fmaxf(l.y, r.y), fmaxf(l.z, r.z), 0.0f);
    }
    __syncthreads();
    if (t == 0 && s_front_n)
        s_front_base = __hip_atomic_fetch_add(frontier_count, s_front_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (is_front) frontier_list[s_front_base + my] = j;
}

// range level k >= 3 from level k - 1 (only built when level 2 alone would leave the top nodes with thousands of
// entries to union: more than 4 M leaves)
__global__ __launch_bounds__(256) void refit_level_kernel(refit_levels_t lv, int k)
{
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= lv.count[k]) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const uint32_t c0 = e << kFanShift, c1 = min(c0 + (1u << kFanShift), lv.count[k - 1]);
    for (uint32_t c = c0; c < c1; c++) {
        float a[3], b[3];
        load_box_plain(&lv.box[k - 1][c], a, b);
#pragma unroll
        for (int d = 0; d < 3; d++) { mn[d] = fminf(mn[d], a[d]); mx[d] = fmaxf(mx[d], b[d]); }
    }
    store_box_plain(&lv.box[k][e], mn, mx);
}

// The leaf range [first, last] of internal node p from the node arrays alone (see the header comment).
__device__ __forceinline__ void node_range(uint32_t n, const lbvh_internal_node* __restrict__ internal, uint32_t p,
                                           uint32_t* first, uint32_t* last)
{
    const uint32_t* nd = reinterpret_cast<const uint32_t*>(&internal[p]);
    const bool first_kind = p <= nd[0];                 // leftNode = split
    uint32_t other = first_kind ? n - 1 : 0u;           // nothing of the other kind up to the root
    uint32_t c = nd[4];                                 // parent
    if (p == 0) c = 0xFFFFFFFFu;                        // the root's parent word is never written (BVH.compute:126,144)
    for (int guard = 0; c < n - 1 && guard < 64; guard++) {
        const uint32_t* cd = reinterpret_cast<const uint32_t*>(&internal[c]);
        if ((c <= cd[0]) != first_kind) { other = c; break; }
        if (c == 0) break;                              // the root is of my kind: the range runs to the end of the array
        c = cd[4];
    }
    *first = first_kind ? p : other;
    *last = first_kind ? other : p;
}

__global__ __launch_bounds__(256) void refit_frontier_kernel(uint32_t n, const lbvh_internal_node* __restrict__ internal,
                                                             const uint32_t* __restrict__ count, const uint32_t* __restrict__ list,
                                                             const lbvh_aabb* __restrict__ tri_aabb,
                                                             const uint32_t* __restrict__ sorted_indices, lbvh_aabb* bvh,
                                                             refit_levels_t lv)
{
    const uint32_t lane = lane_id();
    const uint32_t waves = gridDim.x * 4u;
    const uint32_t total = *count;
    for (uint32_t e = blockIdx.x * 4u + (threadIdx.x >> 6); e < total; e += waves) {
        const uint32_t node = list[e];
        uint32_t first, last;
        node_range(n, internal, node, &first, &last);           // wave-uniform walk
        if (first > last || last >= n) continue;                // malformed caller tree
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        auto take = [&](const lbvh_aabb* p) {
            float a[3], b[3];
            load_box_plain(p, a, b);
#pragma unroll
            for (int d = 0; d < 3; d++) { mn[d] = fminf(mn[d], a[d]); mx[d] = fmaxf(mx[d], b[d]); }
        };
        // [lo, hi) in units of the current level: the ragged ends are taken at this level, the aligned middle
        // moves one level up
        uint32_t lo = first, hi = last + 1u;
        {
            const uint32_t a_end = min(hi, (lo + 63u) & ~63u), b_start = max(a_end, hi & ~63u);
            if (lo + lane < a_end) { const uint32_t i = lo + lane; take(&tri_aabb[sorted_indices ? sorted_indices[i] : i]); }
            if (b_start + lane < hi) { const uint32_t i = b_start + lane; take(&tri_aabb[sorted_indices ? sorted_indices[i] : i]); }
            lo = (lo + 63u) >> kLevel1Shift;
            hi >>= kLevel1Shift;
        }
        for (int k = 1; k <= lv.levels && lo < hi; k++) {
            if (k == lv.levels) {                       // top stored level: everything that is left
                for (uint32_t c = lo + lane; c < hi; c += 64u) take(&lv.box[k][c]);
                break;
            }
            const uint32_t a_end = min(hi, (lo + 15u) & ~15u), b_start = max(a_end, hi & ~15u);
            if (lane < 16u) { if (lo + lane < a_end) take(&lv.box[k][lo + lane]); }
            else if (lane < 32u) { if (b_start + (lane - 16u) < hi) take(&lv.box[k][b_start + (lane - 16u)]); }
            lo = (lo + 15u) >> kFanShift;
            hi >>= kFanShift;
        }
        wave_union(mn, mx);
        if (lane == 0) store_box_plain(&bvh[node], mn, mx);
    }
}

// ---------------------------------------------------------------------------------------------
// Aligned keys for the DERIVED traversal tree (no reference counterpart).  DistributeKeys makes the
// reference's keys unique by accumulating max(diff, 1), which shifts every later key and so cuts the
// Z-order curve at unaligned places (fat, overlapping boxes: 113 node fetches per 16x8 packet on the
// 1 M-triangle scene).  The traversal tree instead perturbs the raw Morton codes as little as
// uniqueness needs: k'_i = i + max_{j<=i}(k_j - j)  (68 fetches per packet, same hit results).
// The raw code of sorted position i is recomputed from the triangle AABB exactly as a-1 does.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t wave_inclusive_max(int32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t o = __shfl_up(v, d);
        if ((int)lane_id() >= d) v = max(v, o);
    }
    return v;
}

constexpr int kAkThreads = 256;

// inclusive prefix max inside the block; returns this thread's inclusive value and the block max
__device__ __forceinline__ int32_t block_inclusive_max(int32_t v, int32_t* s_wave, int32_t* block_max)
{
    const uint32_t t = threadIdx.x;
    const int32_t incl = wave_inclusive_max(v);
    if ((t & 63) == 63) s_wave[t >> 6] = incl;
    __syncthreads();
    int32_t prefix = INT32_MIN, all = INT32_MIN;
#pragma unroll
    for (uint32_t i = 0; i < kAkThreads / LBVH_WAVE; i++) {
        const int32_t x = s_wave[i];
        if (i < (t >> 6)) prefix = max(prefix, x);
        all = max(all, x);
    }
    *block_max = all;
    return max(incl, prefix);
}

// One pass over the sorted order does the three jobs every later kernel of the build starts from: the triangle AABBs
// are gathered into leaf order (the one random gather; level 0 of the range hierarchy), levels 1 .. 10 of the hierarchy
// (unions of aligned groups of 2 .. 1024 leaves) are formed on the way — inside a wave by lane exchange, across the
// workgroup's 16 waves-worth of leaves through LDS — and (TERMS) the aligned-key terms code_i - i are written where the
// derived tree's keys will be, with the maximum of every 1024-term chunk.
constexpr int kAkItems = 4;
constexpr uint32_t kAkChunk = kAkThreads * kAkItems;       // 1024 leaves per workgroup = hierarchy level 10
constexpr uint32_t kHierLocalLevels = 10;

__device__ __forceinline__ void store_hier(const hier_t& h, uint32_t k, uint32_t block, const float mn[3], const float mx[3])
{
    const size_t e = (size_t)hier_offset(h.n2x2, k) + block;
    store_corner(h.lo + e, mn[0], mn[1], mn[2]);
    store_corner(h.hi + e, mx[0], mx[1], mx[2]);
}

template <bool TERMS>
__device__ __forceinline__ void gather_hier_body(uint32_t block, const lbvh_aabb* __restrict__ tri_aabb,
                                                 const uint32_t* __restrict__ sorted_indices, uint32_t n, const box3& scene,
                                                 const hier_t& hier, int32_t* __restrict__ terms, int32_t* __restrict__ chunk_max)
{
    __shared__ int32_t s_wave[kAkThreads / LBVH_WAVE];
    __shared__ float s_six[6][kAkItems * (kAkThreads / LBVH_WAVE)];     // the 16 level-6 boxes of this workgroup, in leaf order
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    int32_t mine = INT32_MIN;
    if (block == 0 && threadIdx.x == 0) {      // the NEUTRAL box of range_boxes: the unused slot behind the top level
        store_corner(hier.lo + (hier.n2x2 - 1u), INFINITY, INFINITY, INFINITY);
        store_corner(hier.hi + (hier.n2x2 - 1u), -INFINITY, -INFINITY, -INFINITY);
    }
#pragma unroll
    for (int k = 0; k < kAkItems; k++) {
        const uint32_t i = block * kAkChunk + (uint32_t)k * kAkThreads + threadIdx.x;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (i < n) {
            const float4* src = reinterpret_cast<const float4*>(&tri_aabb[sorted_indices[i]]);
            const float4 a = src[0], b = src[1];
            mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
            mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
            store_hier(hier, 0, i, mn, mx);                               // level 0: the exact floats BVH.compute:196-205 reads
            if (TERMS) {
                uint32_t q[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    float cen = (mn[d] + mx[d]) * 0.5f;
                    cen = cen - scene.mn[d];
                    cen = cen / (scene.mx[d] - scene.mn[d]);
                    q[d] = quantize(cen);
                }
                const int32_t term = (int32_t)(expand_bits(q[0]) * 4u + expand_bits(q[1]) * 2u + expand_bits(q[2])) - (int32_t)i;
                terms[i] = term;
                mine = max(mine, term);
            }
        }
        // levels 1 .. 6 inside the wave: after step s the lanes whose low s bits are zero hold the union of their 2^s leaves
#pragma unroll
        for (uint32_t lev = 1; lev <= 6; lev++) {
#pragma unroll
            for (int d = 0; d < 3; d++) {
                mn[d] = fminf(mn[d], __shfl_xor(mn[d], 1 << (lev - 1)));
                mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], 1 << (lev - 1)));
            }
            if ((lane & ((1u << lev) - 1u)) == 0 && i < n && lev <= hier.levels) store_hier(hier, lev, i >> lev, mn, mx);
        }
        if (lane == 0) {
#pragma unroll
            for (int d = 0; d < 3; d++) { s_six[d][k * 4 + w] = mn[d]; s_six[3 + d][k * 4 + w] = mx[d]; }
        }
    }
    if (TERMS) {
        int32_t all;
        (void)block_inclusive_max(mine, s_wave, &all);                    // (barrier inside)
        if (threadIdx.x == 0) chunk_max[block] = all;
    } else {
        __syncthreads();
    }
    // levels 7 .. 10 from the 16 level-6 boxes (entry e covers leaves [block * 1024 + 64 e, + 64))
    if (threadIdx.x < 16u) {
        const uint32_t e = threadIdx.x, start = block * kAkChunk + e * 64u;
        float mn[3], mx[3];
#pragma unroll
        for (int d = 0; d < 3; d++) { mn[d] = s_six[d][e]; mx[d] = s_six[3 + d][e]; }
#pragma unroll
        for (uint32_t lev = 7; lev <= kHierLocalLevels; lev++) {
#pragma unroll
            for (int d = 0; d < 3; d++) {
                mn[d] = fminf(mn[d], __shfl_xor(mn[d], 1 << (lev - 7), 16));
                mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], 1 << (lev - 7), 16));
            }
            if ((e & ((1u << (lev - 6)) - 1u)) == 0 && start < n && lev <= hier.levels) store_hier(hier, lev, start >> lev, mn, mx);
        }
    }
}

template <bool TERMS>
__global__ __launch_bounds__(kAkThreads) void gather_hier_kernel(const lbvh_aabb* __restrict__ tri_aabb,
                                                                 const uint32_t* __restrict__ sorted_indices, uint32_t n,
                                                                 box3 scene, hier_t hier, int32_t* __restrict__ terms,
                                                                 int32_t* __restrict__ chunk_max)
{
    gather_hier_body<TERMS>(blockIdx.x, tri_aabb, sorted_indices, n, scene, hier, terms, chunk_max);
}

// hierarchy levels above 10 (blocks of 2048 leaves and more): a few hundred boxes at 1 M leaves, one workgroup.
// Level 10 is read once into LDS (when it fits: up to 4 M leaves) and every higher level is formed there — one barrier
// per level instead of a store -> barrier -> load round trip through memory.
template <uint32_t THREADS, uint32_t LDS_BOXES>
__device__ __forceinline__ void hier_top_levels(const hier_t& hier, uint32_t n, float (&s_box)[6][LDS_BOXES])
{
    constexpr uint32_t kPer = (LDS_BOXES / 2 + THREADS - 1) / THREADS;                   // entries of a level per thread
    uint32_t count = (n + (1u << kHierLocalLevels) - 1u) >> kHierLocalLevels;          // blocks of level 10
    uint32_t lev = kHierLocalLevels;
    // levels too wide for the LDS image go through memory
    while (count > LDS_BOXES) {
        const uint32_t next = (count + 1u) >> 1;
        for (uint32_t e = threadIdx.x; e < next; e += THREADS) {
            const size_t c0 = (size_t)hier_offset(hier.n2x2, lev) + 2 * e;
            f3 a = load_corner(hier.lo + c0), b = load_corner(hier.hi + c0);
            if (2 * e + 1 < count) {
                const f3 a2 = load_corner(hier.lo + c0 + 1), b2 = load_corner(hier.hi + c0 + 1);
                a.x = fminf(a.x, a2.x); a.y = fminf(a.y, a2.y); a.z = fminf(a.z, a2.z);
                b.x = fmaxf(b.x, b2.x); b.y = fmaxf(b.y, b2.y); b.z = fmaxf(b.z, b2.z);
            }
            const float mn[3] = {a.x, a.y, a.z}, mx[3] = {b.x, b.y, b.z};
            store_hier(hier, lev + 1, e, mn, mx);
        }
        __syncthreads();          // the next level reads what this one wrote (same workgroup: visible after the barrier)
        count = next;
        lev++;
    }
    for (uint32_t e = threadIdx.x; e < count; e += THREADS) {
        const size_t c = (size_t)hier_offset(hier.n2x2, lev) + e;
        const f3 a = load_corner(hier.lo + c), b = load_corner(hier.hi + c);
        s_box[0][e] = a.x; s_box[1][e] = a.y; s_box[2][e] = a.z;
        s_box[3][e] = b.x; s_box[4][e] = b.y; s_box[5][e] = b.z;
    }
    __syncthreads();
    while (lev < hier.levels) {
        const uint32_t next = (count + 1u) >> 1;
        float mn[kPer][3], mx[kPer][3];
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            const uint32_t e = threadIdx.x + k * THREADS;
            if (e < next) {
                const bool two = 2 * e + 1 < count;
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    mn[k][d] = two ? fminf(s_box[d][2 * e], s_box[d][2 * e + 1]) : s_box[d][2 * e];
                    mx[k][d] = two ? fmaxf(s_box[3 + d][2 * e], s_box[3 + d][2 * e + 1]) : s_box[3 + d][2 * e];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            const uint32_t e = threadIdx.x + k * THREADS;
            if (e < next) {
#pragma unroll
                for (int d = 0; d < 3; d++) { s_box[d][e] = mn[k][d]; s_box[3 + d][e] = mx[k][d]; }
                store_hier(hier, lev + 1, e, mn[k], mx[k]);
            }
        }
        __syncthreads();
        count = next;
        lev++;
    }
}

constexpr uint32_t kTopLds = 4096;
__global__ __launch_bounds__(1024) void hier_top_kernel(hier_t hier, uint32_t n)
{
    __shared__ float s_box[6][kTopLds];
    hier_top_levels<1024, kTopLds>(hier, n, s_box);
}

__global__ __launch_bounds__(kAkThreads) void aligned_keys_scan_kernel(int32_t* __restrict__ chunk_max, uint32_t chunks)
{
    __shared__ int32_t s_wave[kAkThreads / LBVH_WAVE];
    __shared__ int32_t s_incl[kAkThreads];
    const uint32_t t = threadIdx.x;
    int32_t carry = INT32_MIN;     // max over all chunks of earlier rounds
    for (uint32_t c0 = 0; c0 < chunks; c0 += kAkThreads) {
        const uint32_t c = c0 + t;
        const int32_t x = c < chunks ? chunk_max[c] : INT32_MIN;
        int32_t all;
        s_incl[t] = block_inclusive_max(x, s_wave, &all);
        __syncthreads();
        const int32_t excl = t > 0 ? s_incl[t - 1] : INT32_MIN;
        if (c < chunks) chunk_max[c] = max(carry, excl);      // exclusive prefix max
        carry = max(carry, all);
        __syncthreads();
    }
}

// keys[i] = i + max_{j <= i} term_j, in place over the terms
// SELF: chunk_excl still holds the raw per-chunk maxima (see distribute_apply_kernel)
// One workgroup past the chunks (hier.levels != 0) forms the hierarchy's top levels meanwhile: they depend on the gather
// alone, like this scan — one launch less on the derived lane's critical path (8.5 us + a gap).
constexpr uint32_t kTopLdsSmall = 1024;
template <bool SELF>
__device__ __forceinline__ void aligned_keys_apply_body(uint32_t block, uint32_t n, const int32_t* __restrict__ chunk_excl,
                                                        uint32_t* keys, const hier_t& hier)
{
    __shared__ int32_t s_wave[kAkThreads / LBVH_WAVE];
    __shared__ int32_t s_before[kAkThreads / LBVH_WAVE];
    if (block * kAkChunk >= n) {          // the extra workgroup
        __shared__ float s_box[6][kTopLdsSmall];
        hier_top_levels<kAkThreads, kTopLdsSmall>(hier, n, s_box);      // (measured: 1.4 of this launch's 10.8 us)
        return;
    }
    int32_t carry;
    if (SELF) {
        int32_t part = INT32_MIN;
        for (uint32_t c = threadIdx.x; c < block; c += kAkThreads) part = max(part, chunk_excl[c]);
        (void)block_inclusive_max(part, s_before, &carry);
    } else {
        carry = chunk_excl[block];
    }
#pragma unroll
    for (int k = 0; k < kAkItems; k++) {
        const uint32_t i = block * kAkChunk + (uint32_t)k * kAkThreads + threadIdx.x;
        const int32_t v = i < n ? (int32_t)keys[i] : INT32_MIN;
        int32_t all;
        const int32_t incl = block_inclusive_max(v, s_wave, &all);
        if (i < n) keys[i] = (uint32_t)((int32_t)i + max(incl, carry));
        carry = max(carry, all);
        __syncthreads();           // s_wave is reused by the next sub-chunk
    }
}

template <bool SELF>
__global__ __launch_bounds__(kAkThreads) void aligned_keys_apply_kernel(uint32_t n, const int32_t* __restrict__ chunk_excl,
                                                                        uint32_t* keys, hier_t hier)
{
    aligned_keys_apply_body<SELF>(blockIdx.x, n, chunk_excl, keys, hier);
}

// ---- merged launches of lbvh_build_scene (scenes whose scans fit one workgroup's self-scan: up to 2 M triangles) --------------
// After the sort the build has two independent strands — the derived scene's (gather + hierarchy, aligned keys, derived tree) and
// the reference's (DistributeKeys, reference tree; its boxes come from the same hierarchy).  Rounds 2 - 3 ran them on two
// streams; every cross-stream dependency of the replayed graph cost ~5 us of idle chip (fork after the sort, "hierarchy ready",
// join).  Here each strand's k-th kernel shares ONE launch with the other's: three launches on one stream, the same workgroups,
// the same results.
__global__ __launch_bounds__(256) void gather_and_reduce_kernel(const lbvh_aabb* __restrict__ tri_aabb,
                                                                const uint32_t* __restrict__ sorted_indices, uint32_t n, box3 scene,
                                                                hier_t hier, int32_t* __restrict__ terms, int32_t* __restrict__ chunk_max,
                                                                uint32_t gather_blocks, const uint32_t* __restrict__ keys,
                                                                uint32_t* __restrict__ chunk_sums, uint32_t* __restrict__ boundary)
{
    // the long strand first (the gather's random reads), DistributeKeys' reduction fills in behind it
    if (blockIdx.x < gather_blocks) gather_hier_body<true>(blockIdx.x, tri_aabb, sorted_indices, n, scene, hier, terms, chunk_max);
    else distribute_reduce_body(blockIdx.x - gather_blocks, keys, n, chunk_sums, boundary);
}

__global__ __launch_bounds__(256) void apply_pair_kernel(uint32_t n, const int32_t* __restrict__ chunk_max, uint32_t* aligned_keys,
                                                         hier_t hier, uint32_t aligned_blocks, uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ chunk_sums,
                                                         const uint32_t* __restrict__ boundary)
{
    // block 0: the hierarchy's top levels (one workgroup, the longest of this launch: first), then the aligned keys, then
    // DistributeKeys' apply pass
    if (blockIdx.x < aligned_blocks)
        aligned_keys_apply_body<true>(blockIdx.x == 0 ? aligned_blocks - 1u : blockIdx.x - 1u, n, chunk_max, aligned_keys, hier);
    else
        distribute_apply_body<true>(blockIdx.x - aligned_blocks, keys, n, chunk_sums, boundary);
}

}  // namespace

int lbvh_launch_tree(lbvh_context* ctx, uint32_t n, const uint32_t* d_keys, lbvh_internal_node* d_internal,
                     lbvh_leaf_node* d_leaf, uint32_t* d_zero_word)
{
    const uint32_t blocks = (n - 1 + kTreeThreads - 1) / kTreeThreads;
    LBVH_LAUNCH(ctx, tree_kernel<TREE_TOPOLOGY>, dim3(blocks), dim3(kTreeThreads), d_keys, n, d_internal, d_leaf, d_zero_word,
                hier_t{nullptr, nullptr, 0u, 0u}, (lbvh_aabb*)nullptr, (lbvh_fast_node*)nullptr, 0u, (const uint32_t*)nullptr);
    return LBVH_OK;
}

// the context's range hierarchy for n leaves (grown lazily; one per context: both lanes of a build read the same one)
static int hier_plan(lbvh_context* ctx, uint32_t n, hier_t* h)
{
    uint32_t n2 = 2, levels = 1;
    while (n2 < n) { n2 <<= 1; levels++; }
    const size_t half = ((size_t)2 * n2 * sizeof(corner3) + 255) & ~(size_t)255;
    const int rc = lbvh_reserve(ctx, &ctx->hier, &ctx->hier_bytes, 2 * half);
    if (rc != LBVH_OK) return rc;
    h->lo = (corner3*)ctx->hier;
    h->hi = (corner3*)((char*)ctx->hier + half);
    h->n2x2 = 2u * n2;
    h->levels = levels;
    return LBVH_OK;
}

int lbvh_hier_reserve(lbvh_context* ctx, uint32_t n)
{
    hier_t h;
    return hier_plan(ctx, n, &h);
}

int lbvh_launch_hier_top(lbvh_context* ctx, uint32_t n)
{
    hier_t h;
    const int rc = hier_plan(ctx, n, &h);
    if (rc != LBVH_OK) return rc;
    if (h.levels > kHierLocalLevels) LBVH_LAUNCH(ctx, hier_top_kernel, dim3(1), dim3(1024), h, n);
    return LBVH_OK;
}

int lbvh_launch_gather_hier(lbvh_context* ctx, uint32_t n, const lbvh_aabb* d_triangle_aabb, const uint32_t* d_sorted_indices,
                            const float box_min[3], const float box_max[3], uint32_t* d_aligned_keys_out, bool with_top)
{
    hier_t h;
    int rc = hier_plan(ctx, n, &h);
    if (rc != LBVH_OK) return rc;
    box3 scene;
    for (int k = 0; k < 3; k++) { scene.mn[k] = box_min ? box_min[k] : 0.0f; scene.mx[k] = box_max ? box_max[k] : 1.0f; }
    const uint32_t chunks = (n + kAkChunk - 1) / kAkChunk;
    if (d_aligned_keys_out) {
        rc = lbvh_reserve(ctx, &ctx->scan_scratch[ctx->lane], &ctx->scan_scratch_bytes[ctx->lane], (size_t)chunks * 8);
        if (rc != LBVH_OK) return rc;
        int32_t* chunk_max = (int32_t*)ctx->scan_scratch[ctx->lane];
        LBVH_LAUNCH(ctx, gather_hier_kernel<true>, dim3(chunks), dim3(kAkThreads), d_triangle_aabb, d_sorted_indices, n, scene, h,
                    (int32_t*)d_aligned_keys_out, chunk_max);
        // (with_top: the top levels ride on the apply kernel as one more workgroup)
        const bool ride = with_top && h.levels > kHierLocalLevels;
        if (chunks <= kSelfScanChunks) {
            LBVH_LAUNCH(ctx, aligned_keys_apply_kernel<true>, dim3(chunks + (ride ? 1u : 0u)), dim3(kAkThreads), n, chunk_max,
                        d_aligned_keys_out, h);
        } else {
            LBVH_LAUNCH(ctx, aligned_keys_scan_kernel, dim3(1), dim3(kAkThreads), chunk_max, chunks);
            LBVH_LAUNCH(ctx, aligned_keys_apply_kernel<false>, dim3(chunks + (ride ? 1u : 0u)), dim3(kAkThreads), n, chunk_max,
                        d_aligned_keys_out, h);
        }
        return LBVH_OK;
    } else {
        LBVH_LAUNCH(ctx, gather_hier_kernel<false>, dim3(chunks), dim3(kAkThreads), d_triangle_aabb, d_sorted_indices, n, scene, h,
                    (int32_t*)nullptr, (int32_t*)nullptr);
    }
    if (with_top && h.levels > kHierLocalLevels) LBVH_LAUNCH(ctx, hier_top_kernel, dim3(1), dim3(1024), h, n);
    return LBVH_OK;
}

int lbvh_launch_tree_boxes(lbvh_context* ctx, uint32_t n, const uint32_t* d_keys, lbvh_internal_node* d_internal,
                           lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh)
{
    hier_t h;
    const int rc = hier_plan(ctx, n, &h);
    if (rc != LBVH_OK) return rc;
    const uint32_t blocks = (n - 1 + kTreeThreads - 1) / kTreeThreads;
    LBVH_LAUNCH(ctx, tree_kernel<TREE_REFERENCE>, dim3(blocks), dim3(kTreeThreads), d_keys, n, d_internal, d_leaf, (uint32_t*)nullptr, h,
                d_bvh, (lbvh_fast_node*)nullptr, 0u, (const uint32_t*)nullptr);
    return LBVH_OK;
}

int lbvh_launch_tree_fused(lbvh_context* ctx, uint32_t n, const uint32_t* d_keys, const uint32_t* d_sorted_indices,
                           lbvh_fast_node* d_fused, uint32_t leaf_base)
{
    hier_t h;
    const int rc = hier_plan(ctx, n, &h);
    if (rc != LBVH_OK) return rc;
    const uint32_t blocks = (n - 1 + kTreeThreads - 1) / kTreeThreads;
    LBVH_LAUNCH(ctx, tree_kernel<TREE_FUSED>, dim3(blocks), dim3(kTreeThreads), d_keys, n, (lbvh_internal_node*)nullptr,
                (lbvh_leaf_node*)nullptr, (uint32_t*)nullptr, h, (lbvh_aabb*)nullptr, d_fused, leaf_base, d_sorted_indices);
    return LBVH_OK;
}

// lbvh_build_scene with LBVH_BUILD_FAST_SCENE after the sort, as three merged launches on the current stream (see
// gather_and_reduce_kernel).  *done = false and nothing enqueued when the scene is too large for the self-scanning forms:
// the caller runs the two-stream chain instead.
// does a scene of n triangles take the merged launches (the self-scanning apply passes cover it)?
bool lbvh_post_sort_merges(uint32_t n)
{
    const uint32_t a_chunks = (n + kAkChunk - 1) / kAkChunk;
    const uint32_t d_chunks = (uint32_t)(((uint64_t)n + kDistChunk - 1) / kDistChunk);
    return a_chunks <= kSelfScanChunks && d_chunks <= kSelfScanChunks;
}

int lbvh_launch_post_sort_merged(lbvh_context* ctx, uint32_t n, uint32_t* d_keys, const lbvh_aabb* d_triangle_aabb,
                                 const uint32_t* d_sorted_indices, const float box_min[3], const float box_max[3],
                                 uint32_t* d_aligned_keys, lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh,
                                 lbvh_fast_node* d_fused, uint32_t leaf_base, bool* done)
{
    static_assert(kAkThreads == 256 && kDistThreads == 256 && kTreeThreads == 256, "the merged launches run 256-thread bodies");
    *done = false;
    if (!lbvh_post_sort_merges(n)) return LBVH_OK;
    const uint32_t a_chunks = (n + kAkChunk - 1) / kAkChunk;
    const uint32_t d_chunks = (uint32_t)(((uint64_t)n + kDistChunk - 1) / kDistChunk);
    hier_t h;
    int rc = hier_plan(ctx, n, &h);
    if (rc != LBVH_OK) return rc;
    // the same scratch the two-stream chain uses: lane 1's for the aligned keys' chunk maxima, lane 0's for DistributeKeys
    if ((rc = lbvh_reserve(ctx, &ctx->scan_scratch[1], &ctx->scan_scratch_bytes[1], (size_t)a_chunks * 8)) != LBVH_OK) return rc;
    if ((rc = lbvh_reserve(ctx, &ctx->scan_scratch[0], &ctx->scan_scratch_bytes[0], (size_t)d_chunks * 8)) != LBVH_OK) return rc;
    int32_t* chunk_max = (int32_t*)ctx->scan_scratch[1];
    uint32_t* chunk_sums = (uint32_t*)ctx->scan_scratch[0];
    uint32_t* boundary = chunk_sums + d_chunks;
    box3 scene;
    for (int k = 0; k < 3; k++) { scene.mn[k] = box_min[k]; scene.mx[k] = box_max[k]; }
    LBVH_LAUNCH(ctx, gather_and_reduce_kernel, dim3(a_chunks + d_chunks), dim3(256), d_triangle_aabb, d_sorted_indices, n, scene, h,
                (int32_t*)d_aligned_keys, chunk_max, a_chunks, (const uint32_t*)d_keys, chunk_sums, boundary);
    const uint32_t a_blocks = a_chunks + (h.levels > kHierLocalLevels ? 1u : 0u);       // + the top levels' workgroup
    LBVH_LAUNCH(ctx, apply_pair_kernel, dim3(a_blocks + d_chunks), dim3(256), n, (const int32_t*)chunk_max, d_aligned_keys, h, a_blocks,
                d_keys, (const uint32_t*)chunk_sums, (const uint32_t*)boundary);
    const uint32_t t_blocks = (n - 1 + kTreeThreads - 1) / kTreeThreads;
    LBVH_LAUNCH(ctx, tree_pair_kernel, dim3(2 * t_blocks), dim3(kTreeThreads), (const uint32_t*)d_aligned_keys, (const uint32_t*)d_keys, n,
                d_internal, d_leaf, h, d_bvh, d_fused, leaf_base, d_sorted_indices);
    *done = true;
    return LBVH_OK;
}

int lbvh_launch_morton(lbvh_context* ctx, const lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                       const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                       lbvh_aabb* d_aabb, uint32_t* d_zero, uint32_t zero_words, lbvh_fast_tri* d_lines,
                       lbvh_internal_node* d_reset_internal, lbvh_leaf_node* d_reset_leaf)
{
    box3 scene;
    for (int k = 0; k < 3; k++) { scene.mn[k] = h_box_min[k]; scene.mx[k] = h_box_max[k]; }
    const uint32_t blocks = (capacity + 255) / 256;
    const node_reset reset = {reinterpret_cast<uint32_t*>(d_reset_internal), reinterpret_cast<uint32_t*>(d_reset_leaf)};
    LBVH_LAUNCH(ctx, morton_aabb_kernel, dim3(blocks), dim3(256), d_triangles, n, capacity, scene, d_keys, d_indices, d_aabb,
                d_zero, zero_words, d_lines, reset);
    return LBVH_OK;
}

int lbvh_launch_animate_morton(lbvh_context* ctx, const lbvh_anim& anim, lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                               const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                               lbvh_aabb* d_aabb, uint32_t* d_zero, uint32_t zero_words, lbvh_fast_tri* d_lines,
                               lbvh_internal_node* d_reset_internal, lbvh_leaf_node* d_reset_leaf)
{
    box3 scene;
    for (int k = 0; k < 3; k++) { scene.mn[k] = h_box_min[k]; scene.mx[k] = h_box_max[k]; }
    const uint32_t blocks = (capacity + 255) / 256;
    const node_reset reset = {reinterpret_cast<uint32_t*>(d_reset_internal), reinterpret_cast<uint32_t*>(d_reset_leaf)};
    LBVH_LAUNCH(ctx, animate_morton_kernel, dim3(blocks), dim3(256), anim.rest, anim.body, (const float4*)anim.centres, anim.cos_angle,
                anim.sin_angle, d_triangles, n, capacity, scene, d_keys, d_indices, d_aabb, d_zero, zero_words, d_lines, reset);
    return LBVH_OK;
}

namespace {
struct refit_plan {
    refit_levels_t lv;
    uint32_t* count;
    uint32_t* list;
};
}

// scratch of the current lane: [frontier count (256 B) | frontier list, one u32 per internal node (every node could
// be one) | range levels]
static int refit_prepare(lbvh_context* ctx, uint32_t n, refit_plan* plan)
{
    refit_levels_t lv = {};
    size_t level_bytes = 0;
    {
        uint32_t c = (n + (1u << kLevel1Shift) - 1) >> kLevel1Shift;
        int k = 1;
        for (;; k++) {
            lv.count[k] = c;
            level_bytes += (size_t)c * sizeof(lbvh_aabb);
            if (c <= 1 || k == kMaxLevels) break;
            c = (c + (1u << kFanShift) - 1) >> kFanShift;
        }
        // the top nodes loop over whatever the highest stored level leaves them: stop adding levels once that is
        // at most 4096 entries (64 iterations of a wave)
        while (k > 2 && lv.count[k - 1] <= 4096u) k--;
        lv.levels = k;
        level_bytes = 0;
        for (int i = 1; i <= k; i++) level_bytes += (size_t)lv.count[i] * sizeof(lbvh_aabb);
    }
    const size_t list_bytes = (((size_t)n * 4) + 255) & ~(size_t)255;
    const size_t need = 256 + list_bytes + level_bytes;
    const int L = ctx->lane;
    if (ctx->refit_scratch_words[L] * 4 < need) {
        void* p = ctx->refit_scratch[L];
        size_t have = ctx->refit_scratch_words[L] * 4;
        int rc = lbvh_reserve(ctx, &p, &have, need);
        ctx->refit_scratch[L] = (uint32_t*)p;
        ctx->refit_scratch_words[L] = have / 4;
        if (rc != LBVH_OK) return rc;
    }
    char* base = (char*)ctx->refit_scratch[L];
    plan->count = (uint32_t*)base;
    plan->list = (uint32_t*)(base + 256);
    {
        char* p = base + 256 + list_bytes;
        for (int k = 1; k <= lv.levels; k++) { lv.box[k] = (lbvh_aabb*)p; p += (size_t)lv.count[k] * sizeof(lbvh_aabb); }
    }
    plan->lv = lv;
    return LBVH_OK;
}

int lbvh_refit_counter(lbvh_context* ctx, uint32_t n, uint32_t** d_counter)
{
    refit_plan plan;
    const int rc = refit_prepare(ctx, n, &plan);
    if (rc != LBVH_OK) return rc;
    *d_counter = plan.count;
    return LBVH_OK;
}

int lbvh_launch_refit(lbvh_context* ctx, uint32_t n, const lbvh_internal_node* d_internal, const lbvh_leaf_node* d_leaf,
                      const lbvh_aabb* d_triangle_aabb, const uint32_t* d_sorted_indices, lbvh_aabb* d_bvh, bool counter_cleared)
{
    refit_plan plan;
    {
        const int rc = refit_prepare(ctx, n, &plan);
        if (rc != LBVH_OK) return rc;
    }
    const refit_levels_t lv = plan.lv;
    uint32_t* count = plan.count;
    uint32_t* list = plan.list;
    if (!counter_cleared) LBVH_HIP_TRY(ctx, hipMemsetAsync(count, 0, 256, ctx->cur_stream));
    const uint32_t blocks = (n + kRefitThreads - 1) / kRefitThreads;
    LBVH_LAUNCH(ctx, refit_kernel, dim3(blocks), dim3(kRefitThreads), n, d_internal, d_leaf, d_triangle_aabb, d_sorted_indices, d_bvh,
                lv, count, list);
    if (blocks > 1) {       // a single workgroup finishes the whole tree in LDS
        for (int k = 3; k <= lv.levels; k++)
            LBVH_LAUNCH(ctx, refit_level_kernel, dim3((lv.count[k] + 255) / 256), dim3(256), lv, k);
        uint32_t fblocks = blocks * 4u;                      // frontier nodes are a few per leaf block
        if (fblocks > 2048u) fblocks = 2048u;
        LBVH_LAUNCH(ctx, refit_frontier_kernel, dim3(fblocks), dim3(256), n, d_internal, count, list, d_triangle_aabb,
                    d_sorted_indices, d_bvh, lv);
    }
    return LBVH_OK;
}

#ifdef LBVH_TREE_CLOCK
extern "C" int lbvh_debug_tree_clock(unsigned long long* out32, int reset)
{
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_tree_clock), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tree_clock), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

extern "C" {

lbvh_status lbvh_morton_aabb(lbvh_context* ctx, const lbvh_triangle* d_triangles, uint32_t n,
                             uint32_t capacity, const float h_box_min[3], const float h_box_max[3],
                             uint32_t* d_keys, uint32_t* d_indices, lbvh_aabb* d_aabb)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, capacity >= n);
    LBVH_REQUIRE(ctx, h_box_min != nullptr && h_box_max != nullptr);
    if (capacity == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_keys != nullptr && d_indices != nullptr);
    LBVH_REQUIRE(ctx, n == 0 || (d_triangles != nullptr && d_aabb != nullptr));
    LBVH_REQUIRE(ctx, ((uintptr_t)d_triangles & 15) == 0 && ((uintptr_t)d_aabb & 15) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    lbvh_note_write(ctx, d_keys, (size_t)capacity * 4);
    lbvh_note_write(ctx, d_indices, (size_t)capacity * 4);
    lbvh_note_write(ctx, d_aabb, (size_t)n * sizeof(lbvh_aabb));
    lbvh_launch_morton(ctx, d_triangles, n, capacity, h_box_min, h_box_max, d_keys, d_indices, d_aabb, nullptr, 0, nullptr);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

lbvh_status lbvh_distribute_keys(lbvh_context* ctx, uint32_t* d_keys, uint32_t n)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (n == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_keys != nullptr);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t chunks = (uint32_t)(((uint64_t)n + kDistChunk - 1) / kDistChunk);
    int rc = lbvh_reserve(ctx, &ctx->scan_scratch[ctx->lane], &ctx->scan_scratch_bytes[ctx->lane], (size_t)chunks * 8);
    if (rc != LBVH_OK) return rc;
    uint32_t* chunk_sums = (uint32_t*)ctx->scan_scratch[ctx->lane];
    uint32_t* boundary = chunk_sums + chunks;
    LBVH_LAUNCH(ctx, distribute_reduce_kernel, dim3(chunks), dim3(kDistThreads),
                       d_keys, n, chunk_sums, boundary);
    if (chunks <= kSelfScanChunks) {
        LBVH_LAUNCH(ctx, distribute_apply_kernel<true>, dim3(chunks), dim3(kDistThreads), d_keys, n, chunk_sums, boundary);
    } else {
        LBVH_LAUNCH(ctx, distribute_scan_kernel, dim3(1), dim3(kDistThreads), chunk_sums, chunks);
        LBVH_LAUNCH(ctx, distribute_apply_kernel<false>, dim3(chunks), dim3(kDistThreads), d_keys, n, chunk_sums, boundary);
    }
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

lbvh_status lbvh_build_tree(lbvh_context* ctx, uint32_t n, const uint32_t* d_sorted_keys,
                            lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, n >= 2);            // the reference underflows n - 1 (BVH.compute:101)
    LBVH_REQUIRE(ctx, n <= 0x7FFFFFFFu);
    LBVH_REQUIRE(ctx, d_sorted_keys != nullptr && d_internal != nullptr && d_leaf != nullptr);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_internal & 7) == 0 && ((uintptr_t)d_leaf & 7) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    lbvh_launch_tree(ctx, n, d_sorted_keys, d_internal, d_leaf, nullptr);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

lbvh_status lbvh_refit(lbvh_context* ctx, uint32_t n, const lbvh_internal_node* d_internal,
                       const lbvh_leaf_node* d_leaf, const lbvh_aabb* d_triangle_aabb,
                       const uint32_t* d_sorted_indices, lbvh_aabb* d_bvh)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, n >= 2);
    LBVH_REQUIRE(ctx, d_internal != nullptr && d_leaf != nullptr && d_triangle_aabb != nullptr &&
                          d_sorted_indices != nullptr && d_bvh != nullptr);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_internal & 7) == 0 && ((uintptr_t)d_triangle_aabb & 15) == 0 &&
                          ((uintptr_t)d_bvh & 15) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = lbvh_launch_refit(ctx, n, d_internal, d_leaf, d_triangle_aabb, d_sorted_indices, d_bvh, false);
    if (rc != LBVH_OK) return rc;
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

}  // extern "C"
