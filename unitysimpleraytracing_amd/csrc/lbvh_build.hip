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
// One thread per leaf walks to the root; the second thread to arrive at a node merges the child
// boxes.  The reference has no fence between a thread's box store and the sibling's read
// (BVH.compute:185-215); on MI355X a CU's L1 is never refreshed by other CUs' stores and the 8 XCD
// L2s are not coherent with each other, so the hand-off is explicit:
//   producer: box stored WRITE-THROUGH (8-byte agent-scope stores = global_store_dwordx2 sc1),
//             s_waitcnt vmcnt(0), then the arrival atomic on the parent's flag;
//   consumer: the thread that draws 1 from the flag reads both child boxes with 8-byte agent-scope
//             loads (global_load_dwordx2 sc1: bypass L1, coherent at the device level).
// No L2 write-back / L1 invalidate fences (a release+acquire fence pair per level cost 4 ms at 1 M
// triangles; this form is ~30x faster).  Boxes of leaf children come from the previous kernel
// (triangleAABB), so plain loads are fine there.
// ---------------------------------------------------------------------------------------------
typedef unsigned long long u64;

__device__ __forceinline__ u64 pack2(float a, float b)
{
    return (u64)__float_as_uint(a) | ((u64)__float_as_uint(b) << 32);
}

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

__device__ __forceinline__ void load_box_plain(const lbvh_aabb* __restrict__ p, float mn[3], float mx[3])
{
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1];
    mn[0] = a.x; mn[1] = a.y; mn[2] = a.z;
    mx[0] = b.x; mx[1] = b.y; mx[2] = b.z;
}

__global__ __launch_bounds__(256) void refit_kernel(uint32_t n, const lbvh_internal_node* __restrict__ internal,
                                                    const lbvh_leaf_node* __restrict__ leaf,
                                                    const lbvh_aabb* __restrict__ tri_aabb,
                                                    const uint32_t* __restrict__ sorted_indices,
                                                    lbvh_aabb* bvh, uint32_t* flags)
{
    const uint32_t thread_id = blockIdx.x * blockDim.x + threadIdx.x;
    if (thread_id >= n) return;                                                            // :179
    uint32_t parent = leaf[thread_id].parent;                                              // :181
    for (int guard = 0; parent != 0xFFFFFFFFu && guard < 64; guard++) {                    // :182
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
        else load_box_plain(&tri_aabb[sorted_indices[l.x]], lmn, lmx);
        if (r.y == LBVH_INTERNAL_NODE) load_box_agent(&bvh[r.x], rmn, rmx);                // :206-213
        else load_box_plain(&tri_aabb[sorted_indices[r.x]], rmn, rmx);
        store_box_agent(&bvh[parent], fminf(lmn[0], rmn[0]), fminf(lmn[1], rmn[1]), fminf(lmn[2], rmn[2]),
                        fmaxf(lmx[0], rmx[0]), fmaxf(lmx[1], rmx[1]), fmaxf(lmx[2], rmx[2])); // MergeAABB :152-170
        parent = next;                                                                     // :217
    }
}

}  // namespace

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
    const uint32_t blocks = (n - 1 + 255) / 256;
    LBVH_LAUNCH(ctx, tree_kernel, dim3(blocks), dim3(256), d_sorted_keys, n,
                       d_internal, d_leaf);
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
    if (ctx->refit_flags_words < (size_t)n) {
        void* p = ctx->refit_flags;
        size_t have = ctx->refit_flags_words * 4;
        int rc = lbvh_reserve(ctx, &p, &have, (size_t)n * 4);
        ctx->refit_flags = (uint32_t*)p;
        ctx->refit_flags_words = have / 4;
        if (rc != LBVH_OK) return rc;
    }
    // flags zeroed per build (the reference zeroes them once, Sc/BVHConstructor.cs:41, and so
    // cannot rebuild)
    LBVH_HIP_TRY(ctx, hipMemsetAsync(ctx->refit_flags, 0, (size_t)n * 4, ctx->stream));
    const uint32_t blocks = (n + 255) / 256;
    LBVH_LAUNCH(ctx, refit_kernel, dim3(blocks), dim3(256), n, d_internal, d_leaf,
                       d_triangle_aabb, d_sorted_indices, d_bvh, ctx->refit_flags);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

}  // extern "C"
