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

__global__ __launch_bounds__(256) void morton_aabb_kernel(const lbvh_triangle* __restrict__ tris,
                                                          uint32_t n, uint32_t capacity, box3 scene,
                                                          uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ indices,
                                                          lbvh_aabb* __restrict__ aabb)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= capacity) return;
    if (i >= n) {   // DataBuffer<uint>(.., uint.MaxValue)  MeshBufferContainer.cs:108-109
        keys[i] = 0xFFFFFFFFu;
        indices[i] = 0xFFFFFFFFu;
        return;
    }
    // only the three padded positions (48 of the 128 bytes) are read
    const float4* p = reinterpret_cast<const float4*>(&tris[i]);
    const float4 a = p[0], b = p[1], c = p[2];
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
    float4* o = reinterpret_cast<float4*>(&aabb[i]);
    o[0] = make_float4(mn[0], mn[1], mn[2], 0.0f);
    o[1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
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

__global__ __launch_bounds__(kDistThreads) void distribute_reduce_kernel(
    const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ chunk_sums,
    uint32_t* __restrict__ boundary)
{
    __shared__ uint32_t s_wave[kDistThreads / LBVH_WAVE];
    const uint32_t base = blockIdx.x * (uint32_t)kDistChunk;
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
        chunk_sums[blockIdx.x] = total;
        boundary[blockIdx.x] = prev;   // old key just before this chunk: the apply pass must not
    }                                   // re-read it, the previous chunk overwrites it in place
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

__global__ __launch_bounds__(kDistThreads) void distribute_apply_kernel(
    uint32_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ chunk_sums,
    const uint32_t* __restrict__ boundary)
{
    __shared__ uint32_t s_wave[kDistThreads / LBVH_WAVE];
    const uint32_t base = blockIdx.x * (uint32_t)kDistChunk;
    const uint32_t j0 = base + threadIdx.x * (uint32_t)kDistItems;
    uint32_t prev = 0;
    if (threadIdx.x == 0) prev = boundary[blockIdx.x];
    else if (j0 - 1 < n) prev = keys[j0 - 1];
    uint32_t f[kDistItems];
    dist_load(keys, n, j0, prev, f);
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < kDistItems; c++) { s += f[c]; f[c] = s; }   // thread-local inclusive
    uint32_t total;
    const uint32_t excl = block_exclusive_sum(s, s_wave, &total);  // barrier inside: every old key
    const uint32_t carry = chunk_sums[blockIdx.x] + excl;          // of the chunk is loaded by now
#pragma unroll
    for (int c = 0; c < kDistItems; c++) {
        const uint32_t j = j0 + (uint32_t)c;
        if (j < n) keys[j] = carry + f[c];
    }
}

// ---------------------------------------------------------------------------------------------
// a-7  TreeConstructor (Karras 2012)          BVH.compute:18-149
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }   // :18-21

__device__ __forceinline__ int delta(const uint32_t* __restrict__ codes, int x_code_idx, uint32_t x_code,
                                     int y, int num_objects)                              // :23-33
{
    (void)x_code_idx;
    if (y >= 0 && y <= num_objects - 1) return clz32(x_code ^ codes[y]);
    return -1;
}

__global__ __launch_bounds__(256) void tree_kernel(const uint32_t* __restrict__ codes, uint32_t n,
                                                   lbvh_internal_node* __restrict__ internal,
                                                   lbvh_leaf_node* __restrict__ leaf)
{
    const uint32_t thread_id = blockIdx.x * blockDim.x + threadIdx.x;
    if (thread_id >= n - 1) return;                                                        // :101
    const int num = (int)n;
    const int idx = (int)thread_id;
    const uint32_t self = codes[idx];

    // DetermineRange :35-52 (idx is always in range, so delta only range-checks the other end)
    const int dl = delta(codes, idx, self, idx - 1, num);
    const int dr = delta(codes, idx, self, idx + 1, num);
    const int diff = dr - dl;
    const int d = (diff > 0) - (diff < 0);                                                 // sign(), :37
    const int dmin = d > 0 ? dl : (d < 0 ? dr : clz32(0u));                                // :38
    uint32_t lmax = 2;                                                                     // :39
    while (delta(codes, idx, self, idx + (int)(lmax * (uint32_t)d), num) > dmin) lmax *= 2; // :40-41
    int l = 0;
    for (uint32_t t = lmax / 2; t >= 1; t /= 2) {                                          // :43
        if (delta(codes, idx, self, idx + (int)(((uint32_t)l + t) * (uint32_t)d), num) > dmin)
            l += (int)t;                                                                   // :45-46
    }
    const int j = idx + l * d;                                                             // :49
    const int first = min(idx, j), last = max(idx, j);                                     // :50

    // FindSplit :54-92
    int split;
    {
        const uint32_t first_code = codes[first];
        const uint32_t last_code = codes[last];
        if (first_code == last_code) {
            split = (first + last) >> 1;                                                   // :61-62
        } else {
            const int common_prefix = clz32(first_code ^ last_code);                       // :67
            split = first;
            int step = last - first;
            do {
                step = (step + 1) >> 1;                                                    // :78
                const int new_split = split + step;
                if (new_split < last) {
                    const int split_prefix = clz32(first_code ^ codes[new_split]);
                    if (split_prefix > common_prefix) split = new_split;                   // :85-86
                }
            } while (step > 1);
        }
    }
    if (split < 0 || (uint32_t)split + 1u >= n) return;   // only reachable with non-unique keys

    const bool left_leaf = split == first;                                                 // :114
    const bool right_leaf = split + 1 == last;                                             // :132
    uint32_t* node = reinterpret_cast<uint32_t*>(&internal[thread_id]);
    // leftNode, leftNodeType, rightNode, rightNodeType as two 8-byte stores (node stride 24 B)
    *reinterpret_cast<uint2*>(node + 0) = make_uint2((uint32_t)split, left_leaf ? LBVH_LEAF_NODE : LBVH_INTERNAL_NODE);
    *reinterpret_cast<uint2*>(node + 2) = make_uint2((uint32_t)split + 1u, right_leaf ? LBVH_LEAF_NODE : LBVH_INTERNAL_NODE);
    node[5] = thread_id;                                                                   // index :111
    if (left_leaf) *reinterpret_cast<uint2*>(&leaf[split]) = make_uint2(thread_id, (uint32_t)split);  // :116-120
    else internal[split].parent = thread_id;                                               // :126
    if (right_leaf) *reinterpret_cast<uint2*>(&leaf[split + 1]) = make_uint2(thread_id, (uint32_t)split + 1u);
    else internal[split + 1].parent = thread_id;                                           // :144
}

// ---------------------------------------------------------------------------------------------
// a-8  BVHConstructor (bottom-up refit)       BVH.compute:152-220
// The reference: one thread per leaf walks to the root; the second thread to arrive at a node (per-node
// counter, InterlockedCompareExchange) merges the child boxes — with no fence between a thread's box
// store and the sibling's read (BVH.compute:185-215).  On MI355X a CU's L1 is never refreshed by other
// CUs' stores and the 8 XCD L2s are not coherent with each other, so that hand-off must not cross
// workgroups unprotected.  Here it never crosses workgroups at all:
typedef unsigned long long u64;

__device__ __forceinline__ u64 pack2(float a, float b)
{
    return (u64)__float_as_uint(a) | ((u64)__float_as_uint(b) << 32);
}

// 8-byte agent-scope stores/loads = global_store/load_dwordx2 sc1: write-through / L1-bypassing
__device__ __forceinline__ void store_box_agent(lbvh_aabb* p, float mnx, float mny, float mnz, float mxx,
                                                float mxy, float mxz)
{
    u64* q = reinterpret_cast<u64*>(p);
    __hip_atomic_store(q + 0, pack2(mnx, mny), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, pack2(mnz, 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 2, pack2(mxx, mxy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 3, pack2(mxz, 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void load_box_agent(const lbvh_aabb* p, float mn[3], float mx[3])
{
    u64* q = reinterpret_cast<u64*>(const_cast<lbvh_aabb*>(p));
    const u64 a = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 c = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 d = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mn[0] = __uint_as_float((uint32_t)a); mn[1] = __uint_as_float((uint32_t)(a >> 32));
    mn[2] = __uint_as_float((uint32_t)b);
    mx[0] = __uint_as_float((uint32_t)c); mx[1] = __uint_as_float((uint32_t)(c >> 32));
    mx[2] = __uint_as_float((uint32_t)d);
}

__device__ __forceinline__ void load_box_plain(const lbvh_aabb* p, float mn[3], float mx[3])
{
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1];
    mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
    mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
}

// One kernel, two phases per 1024-leaf workgroup.
//  Phase 1 (LDS): a thread carries its subtree's box AND leaf range [first, last] upward.  An arrival at
//    parent q is LOCAL when q's index and the carried range lie inside the workgroup's 1024 indices:
//    the thread parks its box in the LDS slot (q, side) and bumps an LDS counter that encodes which
//    side arrived; the second arriver finds the sibling's box and range in LDS, merges, writes bvhData[q]
//    with a plain store and goes on.  No global atomics, no sc1 traffic: > 95 % of all merges.
//  Phase 2 (global): what is left is the frontier — threads whose next arrival is not local, and LDS
//    slots that received only one child (the sibling subtree lives in another workgroup).  Those
//    arrivals cross workgroups, so they use an explicit protocol: the child's box is (re)stored
//    WRITE-THROUGH (8-byte agent-scope stores = global_store_dwordx2 sc1), s_waitcnt vmcnt(0), then the
//    agent-scope arrival counter; the second arriver reads both child boxes with sc1 loads (L1 bypass).
//    No L2 write-back / L1 invalidate fences: a release+acquire fence pair per level measured 3.99 ms
//    for the whole tree, the sc1 form alone 0.18 ms, this hybrid 0.13 ms.  (Replaying the frontier in a
//    second single-workgroup kernel measured slower: the frontier is tens of thousands of subtrees.)
// The rule "local iff carried range and q inside the workgroup" is evaluated identically by whichever
// thread arrives, so the LDS and global counters never disagree about who is second.
constexpr int kRefitThreads = 1024;

__device__ __forceinline__ void refit_global_walk(uint32_t n, const lbvh_internal_node* __restrict__ internal,
                                                  const lbvh_aabb* __restrict__ tri_aabb,
                                                  const uint32_t* __restrict__ sorted_indices, lbvh_aabb* bvh,
                                                  uint32_t* flags, uint32_t parent, bool child_internal, uint32_t child_id,
                                                  const float cmn[3], const float cmx[3])
{
    // the child this thread brings must be readable by a merger on another XCD
    if (child_internal) store_box_agent(&bvh[child_id], cmn[0], cmn[1], cmn[2], cmx[0], cmx[1], cmx[2]);
    for (int guard = 0; parent != 0xFFFFFFFFu && guard < 64; guard++) {                    // BVH.compute:182
        if (parent >= n - 1) break;
        // every box store of this thread has completed before it signals the parent
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(&flags[parent], 1u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);             // :185
        if (old == 0) break;                                                               // :186-189
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler-only: loads stay below

        const uint32_t* nd = reinterpret_cast<const uint32_t*>(&internal[parent]);
        const uint2 l = *reinterpret_cast<const uint2*>(nd + 0);
        const uint2 r = *reinterpret_cast<const uint2*>(nd + 2);
        const uint32_t next = nd[4];
        float lmn[3], lmx[3], rmn[3], rmx[3];
        if (l.y == LBVH_INTERNAL_NODE) load_box_agent(&bvh[l.x], lmn, lmx);                // :197-204
        else load_box_plain(&tri_aabb[sorted_indices ? sorted_indices[l.x] : l.x], lmn, lmx);
        if (r.y == LBVH_INTERNAL_NODE) load_box_agent(&bvh[r.x], rmn, rmx);                // :206-213
        else load_box_plain(&tri_aabb[sorted_indices ? sorted_indices[r.x] : r.x], rmn, rmx);
        store_box_agent(&bvh[parent], fminf(lmn[0], rmn[0]), fminf(lmn[1], rmn[1]), fminf(lmn[2], rmn[2]),
                        fmaxf(lmx[0], rmx[0]), fmaxf(lmx[1], rmx[1]), fmaxf(lmx[2], rmx[2])); // MergeAABB :152-170
        parent = next;                                                                     // :217
    }
}

// sorted_indices == nullptr: tri_aabb is already in sorted (leaf) order
__global__ __launch_bounds__(kRefitThreads) void refit_kernel(uint32_t n, const lbvh_internal_node* __restrict__ internal,
                                                              const lbvh_leaf_node* __restrict__ leaf,
                                                              const lbvh_aabb* __restrict__ tri_aabb,
                                                              const uint32_t* __restrict__ sorted_indices,
                                                              lbvh_aabb* bvh, uint32_t* flags)
{
    __shared__ float s_box[2][6][kRefitThreads];        // [side][min xyz, max xyz][slot]: 48 KB
    __shared__ uint32_t s_range[2][kRefitThreads];      // first | last << 16, relative to the workgroup: 8 KB
    __shared__ uint32_t s_flag[kRefitThreads];          // +1 = left child arrived, +0x10000 = right child
    const uint32_t t = threadIdx.x;
    const uint32_t b0 = blockIdx.x * (uint32_t)kRefitThreads;
    s_flag[t] = 0;
    __syncthreads();

    // ---- phase 1 ----
    const uint32_t j = b0 + t;
    bool pending = false;
    uint32_t q = 0xFFFFFFFFu, child_id = j, first = j, last = j;
    bool child_internal = false;
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    if (j < n) {                                                                           // :179
        load_box_plain(&tri_aabb[sorted_indices ? sorted_indices[j] : j], mn, mx);
        q = leaf[j].parent;                                                                // :181
        for (int guard = 0; q != 0xFFFFFFFFu && guard < 64; guard++) {
            if (q >= n - 1) { q = 0xFFFFFFFFu; break; }
            const bool local = q >= b0 && q - b0 < (uint32_t)kRefitThreads && first >= b0 &&
                               last - b0 < (uint32_t)kRefitThreads;
            if (!local) { pending = true; break; }
            const uint32_t slot = q - b0;
            const uint32_t* nd = reinterpret_cast<const uint32_t*>(&internal[q]);
            const uint2 l = *reinterpret_cast<const uint2*>(nd + 0);
            const uint32_t next = nd[4];
            const uint32_t side = (l.x == child_id && (l.y == LBVH_INTERNAL_NODE) == child_internal) ? 0u : 1u;
            // park my box and range, THEN arrive (LDS executes in issue order: the sibling that draws
            // the second ticket finds them)
#pragma unroll
            for (int k = 0; k < 3; k++) { s_box[side][k][slot] = mn[k]; s_box[side][3 + k][slot] = mx[k]; }
            s_range[side][slot] = (first - b0) | ((last - b0) << 16);
            const uint32_t old = atomicAdd(&s_flag[slot], side == 0 ? 1u : 0x10000u);      // :185 (LDS)
            if (old == 0) { q = 0xFFFFFFFFu; break; }                                      // first arrival :186-189
            const uint32_t o = side ^ 1u;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                mn[k] = fminf(mn[k], s_box[o][k][slot]);                                   // MergeAABB :152-170
                mx[k] = fmaxf(mx[k], s_box[o][3 + k][slot]);
            }
            const uint32_t orange = s_range[o][slot];
            first = min(first, b0 + (orange & 0xFFFFu));
            last = max(last, b0 + (orange >> 16));
            float4* ob = reinterpret_cast<float4*>(&bvh[q]);                               // :215
            ob[0] = make_float4(mn[0], mn[1], mn[2], 0.0f);
            ob[1] = make_float4(mx[0], mx[1], mx[2], 0.0f);
            child_id = q;
            child_internal = true;
            q = next;                                                                      // :217
        }
    }
    __syncthreads();

    // ---- phase 2 ----
    if (pending) refit_global_walk(n, internal, tri_aabb, sorted_indices, bvh, flags, q, child_internal, child_id, mn, mx);
    // LDS slots that saw exactly one child: its sibling subtree belongs to another workgroup
    const uint32_t f = s_flag[t];
    if (f == 1u || f == 0x10000u) {
        const uint32_t side = f == 1u ? 0u : 1u;
        const uint32_t node = b0 + t;
        const uint32_t* nd = reinterpret_cast<const uint32_t*>(&internal[node]);
        const uint2 c = *reinterpret_cast<const uint2*>(nd + 2 * side);
        float cmn[3], cmx[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { cmn[k] = s_box[side][k][t]; cmx[k] = s_box[side][3 + k][t]; }
        refit_global_walk(n, internal, tri_aabb, sorted_indices, bvh, flags, node, c.y == LBVH_INTERNAL_NODE, c.x, cmn, cmx);
    }
}

// out[i] = in[index[i]] for 32-byte boxes: the one random gather of the traversal-tree build
__global__ __launch_bounds__(256) void gather_aabb_kernel(const lbvh_aabb* __restrict__ in, const uint32_t* __restrict__ index,
                                                          uint32_t n, lbvh_aabb* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* src = reinterpret_cast<const float4*>(&in[index[i]]);
    const float4 a = src[0], b = src[1];
    float4* dst = reinterpret_cast<float4*>(&out[i]);
    dst[0] = a;
    dst[1] = b;
}

// ---------------------------------------------------------------------------------------------
// Aligned keys for the DERIVED traversal tree (no reference counterpart).  DistributeKeys makes the
// reference's keys unique by accumulating max(diff, 1), which shifts every later key and so cuts the
// Z-order curve at unaligned places (fat, overlapping boxes: 113 node fetches per 16x8 packet on the
// 1 M-triangle scene).  The traversal tree instead perturbs the raw Morton codes as little as
// uniqueness needs: k'_i = i + max_{j<=i}(k_j - j)  (68 fetches per packet, same hit results).
// The raw code of sorted position i is recomputed from the triangle AABB exactly as a-1 does.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t aligned_term(const lbvh_aabb* __restrict__ tri_aabb,
                                                const uint32_t* __restrict__ sorted_indices, uint32_t i, const box3& scene)
{
    const float4* b = reinterpret_cast<const float4*>(&tri_aabb[sorted_indices ? sorted_indices[i] : i]);
    const float4 mn = b[0], mx = b[1];
    const float bmn[3] = {mn.x, mn.y, mn.z}, bmx[3] = {mx.x, mx.y, mx.z};
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float cen = (bmn[k] + bmx[k]) * 0.5f;
        cen = cen - scene.mn[k];
        cen = cen / (scene.mx[k] - scene.mn[k]);
        q[k] = quantize(cen);
    }
    const uint32_t code = expand_bits(q[0]) * 4u + expand_bits(q[1]) * 2u + expand_bits(q[2]);
    return (int32_t)code - (int32_t)i;          // code < 2^30, i < 2^31
}

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

__global__ __launch_bounds__(kAkThreads) void aligned_keys_reduce_kernel(const lbvh_aabb* __restrict__ tri_aabb,
                                                                         const uint32_t* __restrict__ sorted_indices,
                                                                         uint32_t n, box3 scene, int32_t* __restrict__ chunk_max)
{
    __shared__ int32_t s_wave[kAkThreads / LBVH_WAVE];
    const uint32_t i = blockIdx.x * kAkThreads + threadIdx.x;
    const int32_t v = i < n ? aligned_term(tri_aabb, sorted_indices, i, scene) : INT32_MIN;
    int32_t all;
    (void)block_inclusive_max(v, s_wave, &all);
    if (threadIdx.x == 0) chunk_max[blockIdx.x] = all;
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

__global__ __launch_bounds__(kAkThreads) void aligned_keys_apply_kernel(const lbvh_aabb* __restrict__ tri_aabb,
                                                                        const uint32_t* __restrict__ sorted_indices,
                                                                        uint32_t n, box3 scene,
                                                                        const int32_t* __restrict__ chunk_excl,
                                                                        uint32_t* __restrict__ keys_out)
{
    __shared__ int32_t s_wave[kAkThreads / LBVH_WAVE];
    const uint32_t i = blockIdx.x * kAkThreads + threadIdx.x;
    const int32_t v = i < n ? aligned_term(tri_aabb, sorted_indices, i, scene) : INT32_MIN;
    int32_t all;
    const int32_t incl = block_inclusive_max(v, s_wave, &all);
    if (i < n) keys_out[i] = (uint32_t)((int32_t)i + max(incl, chunk_excl[blockIdx.x]));
}

}  // namespace

int lbvh_launch_tree(lbvh_context* ctx, uint32_t n, const uint32_t* d_keys, lbvh_internal_node* d_internal,
                     lbvh_leaf_node* d_leaf)
{
    const uint32_t blocks = (n - 1 + 255) / 256;
    LBVH_LAUNCH(ctx, tree_kernel, dim3(blocks), dim3(256), d_keys, n, d_internal, d_leaf);
    return LBVH_OK;
}

int lbvh_launch_refit(lbvh_context* ctx, uint32_t n, const lbvh_internal_node* d_internal, const lbvh_leaf_node* d_leaf,
                      const lbvh_aabb* d_triangle_aabb, const uint32_t* d_sorted_indices, lbvh_aabb* d_bvh)
{
    if (ctx->refit_flags_words < (size_t)n) {
        void* p = ctx->refit_flags;
        size_t have = ctx->refit_flags_words * 4;
        int rc = lbvh_reserve(ctx, &p, &have, (size_t)n * 4);
        ctx->refit_flags = (uint32_t*)p;
        ctx->refit_flags_words = have / 4;
        if (rc != LBVH_OK) return rc;
    }
    // counters zeroed per build (the reference zeroes its atomicsData once, Sc/BVHConstructor.cs:41, and
    // so cannot rebuild)
    LBVH_HIP_TRY(ctx, hipMemsetAsync(ctx->refit_flags, 0, (size_t)n * 4, ctx->stream));
    const uint32_t blocks = (n + kRefitThreads - 1) / kRefitThreads;
    LBVH_LAUNCH(ctx, refit_kernel, dim3(blocks), dim3(kRefitThreads), n, d_internal, d_leaf, d_triangle_aabb,
                d_sorted_indices, d_bvh, ctx->refit_flags);
    return LBVH_OK;
}

int lbvh_launch_gather_aabb(lbvh_context* ctx, uint32_t n, const lbvh_aabb* d_in, const uint32_t* d_index, lbvh_aabb* d_out)
{
    LBVH_LAUNCH(ctx, gather_aabb_kernel, dim3((n + 255) / 256), dim3(256), d_in, d_index, n, d_out);
    return LBVH_OK;
}

int lbvh_launch_aligned_keys(lbvh_context* ctx, uint32_t n, const lbvh_aabb* d_triangle_aabb,
                             const uint32_t* d_sorted_indices, const float box_min[3], const float box_max[3],
                             uint32_t* d_keys_out)
{
    box3 scene;
    for (int k = 0; k < 3; k++) { scene.mn[k] = box_min[k]; scene.mx[k] = box_max[k]; }
    const uint32_t chunks = (n + kAkThreads - 1) / kAkThreads;
    int rc = lbvh_reserve(ctx, &ctx->scan_scratch, &ctx->scan_scratch_bytes, (size_t)chunks * 8);
    if (rc != LBVH_OK) return rc;
    int32_t* chunk_max = (int32_t*)ctx->scan_scratch;
    LBVH_LAUNCH(ctx, aligned_keys_reduce_kernel, dim3(chunks), dim3(kAkThreads), d_triangle_aabb, d_sorted_indices, n, scene,
                chunk_max);
    LBVH_LAUNCH(ctx, aligned_keys_scan_kernel, dim3(1), dim3(kAkThreads), chunk_max, chunks);
    LBVH_LAUNCH(ctx, aligned_keys_apply_kernel, dim3(chunks), dim3(kAkThreads), d_triangle_aabb, d_sorted_indices, n, scene,
                chunk_max, d_keys_out);
    return LBVH_OK;
}

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
    box3 scene;
    for (int k = 0; k < 3; k++) { scene.mn[k] = h_box_min[k]; scene.mx[k] = h_box_max[k]; }
    const uint32_t blocks = (capacity + 255) / 256;
    LBVH_LAUNCH(ctx, morton_aabb_kernel, dim3(blocks), dim3(256), d_triangles, n,
                       capacity, scene, d_keys, d_indices, d_aabb);
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
    int rc = lbvh_reserve(ctx, &ctx->scan_scratch, &ctx->scan_scratch_bytes, (size_t)chunks * 8);
    if (rc != LBVH_OK) return rc;
    uint32_t* chunk_sums = (uint32_t*)ctx->scan_scratch;
    uint32_t* boundary = chunk_sums + chunks;
    LBVH_LAUNCH(ctx, distribute_reduce_kernel, dim3(chunks), dim3(kDistThreads),
                       d_keys, n, chunk_sums, boundary);
    LBVH_LAUNCH(ctx, distribute_scan_kernel, dim3(1), dim3(kDistThreads), chunk_sums,
                       chunks);
    LBVH_LAUNCH(ctx, distribute_apply_kernel, dim3(chunks), dim3(kDistThreads),
                       d_keys, n, chunk_sums, boundary);
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
    lbvh_launch_tree(ctx, n, d_sorted_keys, d_internal, d_leaf);
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
    int rc = lbvh_launch_refit(ctx, n, d_internal, d_leaf, d_triangle_aabb, d_sorted_indices, d_bvh);
    if (rc != LBVH_OK) return rc;
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

}  // extern "C"
