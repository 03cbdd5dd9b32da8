// lbvh_trace.hip — primary-ray generation + BVH traversal for gfx950.
//
// Reference: kernel Raytracing, Assets/_Shaders/Raytracing/Raytracing.compute:105-185 (ray gen
// :108-126, traversal :133-176) with RayBoxIntersection :75-87, CheckTriangle :89-103 and
// RayTriangleIntersection :37-73, dispatched by Assets/_Scripts/RaytracingMeshDrawer.cs:76-84.
//
// Two traversal flavours share the ray generation and the exact (strict fp32, no FMA contraction)
// slab and Moeller-Trumbore arithmetic, so the winning t is bit-identical to the CPU oracle:
//   LBVH_TRACE_REFERENCE  walks the reference's own arrays in the reference's visit order (push
//                         left, push right, pop right first; no pruning) — the parity mode and the
//                         source of the reference-semantics visit counters.
//   LBVH_TRACE_FAST       walks the derived 64-byte fused nodes (both child boxes + child refs in
//                         one fetch, the triangles as 64-byte lines in the caller's order behind them), visits
//                         the nearer child first and skips boxes that start beyond the best hit.
//                         Same candidate set minus boxes that cannot win => same min t.
//
// LBVH_TRACE_REFERENCE: one wave (8x8 pixels) per workgroup, per-lane stack in LDS as [entry][lane]
// (bank-conflict-free, no scratch memory), no barriers.  LBVH_TRACE_FAST: one wave per 8x8-pixel packet with
// a wave-shared stack in one VGPR (v_writelane / v_readlane), no LDS at all (see the packet section below).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "lbvh_common.h"
#include "lbvh_rt.h"

namespace {

constexpr int kStackDepth = 34;   // distributed keys are < 2^31 => <= 31 internal levels below the
                                  // root; the reference's push-both order needs depth + 1 entries

struct trace_args {
    lbvh_camera cam;
    int32_t x0, y0, x1, y1;
    uint32_t tiles_x, tiles_y;
    uint32_t shard_index, shard_count;   // this launch traces every shard_count-th GROUP of tiles
    uint32_t line_bytes;                 // size of the node + triangle line array if it is below 4 GB, else 0
    uint32_t packed;                     // 1: records go to hits[work item * 64 + lane] (lbvh_trace_primary_shard_packed)
    // LBVH_TRACE_FAST_EXACT (nullptr otherwise): every triangle a ray meets at exactly its best t is listed here — [0] = count,
    // then entries {record slot, px | py << 16, leaf position, t} — and resolve_ties_* give the record to the one the
    // REFERENCE's visit order meets first
    uint32_t* ties;                      // (words 0 / 1: this frame's count / the next one's — they take turns, and a frame's last
    uint32_t tie_capacity;               // resolve kernel clears the other one: no fill in front of the trace)
    uint32_t tie_turn;
    uint32_t centre_first;               // 1: without history a whole frame's tiles are taken centre-out (trace_packet_kernel)
    const uint32_t* sorted_indices;      // leaf position -> triangle
};

constexpr int kOrderClasses = 16;         // cost classes of the packet dispatch order
constexpr uint32_t kShardGroup = 8;      // adjacent tiles (one strip of the frame) that stay together

// k-th work item of this shard -> tile of the rectangle (row-major); identity for shard_count == 1
__device__ __host__ __forceinline__ uint32_t shard_tile(uint32_t k, uint32_t shard_index, uint32_t shard_count)
{
    return ((k / kShardGroup) * shard_count + shard_index) * kShardGroup + k % kShardGroup;
}

// number of work items (tile slots, the last group may run past n_tiles) owned by a shard
static inline uint32_t shard_work(uint32_t n_tiles, uint32_t shard_index, uint32_t shard_count)
{
    const uint32_t groups = (n_tiles + kShardGroup - 1) / kShardGroup;
    const uint32_t owned = groups > shard_index ? (groups - shard_index + shard_count - 1) / shard_count : 0;
    return owned * kShardGroup;
}

// Workgroup id -> 8x8 pixel tile.  Workgroups are dealt round-robin over the 8 XCDs, so ids
// b, b+8, b+16, ... share one XCD (and its 4-MiB L2); give each XCD a contiguous run of tiles in
// row-major screen order so the BVH subtrees one L2 sees are a compact part of the scene.
// Speed only: any mapping is correct.  Bijective for every tile count.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t n)
{
    const uint32_t per = n / 8, rem = n % 8;
    const uint32_t x = b % 8, k = b / 8;
    // XCD x owns `per` tiles (+1 for the first `rem` XCDs)
    const uint32_t start = x * per + (x < rem ? x : rem);
    return start + k;
}

__device__ __forceinline__ bool tile_pixel(const trace_args& a, uint32_t tile, uint32_t lane, uint32_t& px,
                                           uint32_t& py)
{
    const uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    px = (uint32_t)a.x0 + tx * 8 + (lane & 7);
    py = (uint32_t)a.y0 + ty * 8 + (lane >> 3);
    return px < (uint32_t)a.x1 && py < (uint32_t)a.y1;
}

// where the record of lane `lane` (pixel px, py) of work item w goes: at its pixel of the traced rectangle, or — packed
// shares, one contiguous block per GPU for whatever carries it to the frame's owner — behind the share's earlier tiles
__device__ __forceinline__ size_t hit_slot(const trace_args& a, uint32_t w, uint32_t lane, uint32_t px, uint32_t py)
{
    if (a.packed) return (size_t)w * 64u + lane;
    return (size_t)(py - (uint32_t)a.y0) * (uint32_t)(a.x1 - a.x0) + (px - (uint32_t)a.x0);
}

__device__ __forceinline__ void add_stats(lbvh_trace_stats* stats, uint32_t pops, uint32_t box_hits,
                                          uint32_t leaf_tests, uint32_t tri_tests, uint32_t hits)
{
    // wave reduction, one atomic per counter per wave
    uint32_t v[5] = {pops, box_hits, leaf_tests, tri_tests, hits};
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint32_t incl = wave_inclusive_sum(v[i]);
        v[i] = wave_total_from_inclusive(incl);
    }
    if (lane_id() == 0) {
        atomicAdd((unsigned long long*)&stats->pops, (unsigned long long)v[0]);
        atomicAdd((unsigned long long*)&stats->box_hits, (unsigned long long)v[1]);
        atomicAdd((unsigned long long*)&stats->leaf_tests, (unsigned long long)v[2]);
        atomicAdd((unsigned long long*)&stats->tri_tests, (unsigned long long)v[3]);
        atomicAdd((unsigned long long*)&stats->hits, (unsigned long long)v[4]);
    }
}

// ---------------------------------------------------------------------------------------------
// LBVH_TRACE_REFERENCE
// ---------------------------------------------------------------------------------------------
// one ray through the reference's loop (Raytracing.compute:128-176): the lane's column of s_stack is its stack
template <bool STATS>
__device__ __forceinline__ float4 reference_ray(const lbvh_scene& s, const ray_t& ray, uint32_t (*s_stack)[LBVH_WAVE], uint32_t lane,
                                                uint32_t& n_pops, uint32_t& n_box, uint32_t& n_leaf, uint32_t& n_tri)
{
    float best_t = LBVH_MAX_FLOAT;                    // :129
    uint32_t best_tri = 0;                            // :130
    float best_u = 0.0f, best_v = 0.0f;               // :131

    uint32_t sp = 0;
    s_stack[0][lane] = 0;                             // :135
    sp = 1;
    while (sp != 0) {                                 // :138
        sp--;
        const uint32_t index = s_stack[sp][lane];     // :141
        if (STATS) n_pops++;
        const float4* nb = reinterpret_cast<const float4*>(&s.bvh[index]);
        float tmin;
        if (!ray_box(nb[0], nb[1], ray, tmin)) continue;     // :143-146
        if (STATS) n_box++;
        const uint32_t* nd = reinterpret_cast<const uint32_t*>(&s.internal_nodes[index]);
        const uint2 lc = *reinterpret_cast<const uint2*>(nd + 0);
        const uint2 rc = *reinterpret_cast<const uint2*>(nd + 2);
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const uint2 c = side == 0 ? lc : rc;      // left then right  :148-175
            if (c.y == LBVH_INTERNAL_NODE) {
                if (sp < (uint32_t)kStackDepth) s_stack[sp][lane] = c.x;
                sp++;
            } else {
                const uint32_t tri = s.sorted_indices[s.leaf_nodes[c.x].index];   // :158
                if (STATS) n_leaf++;
                const float4* tb = reinterpret_cast<const float4*>(&s.triangle_aabb[tri]);
                float tmin2;
                if (ray_box(tb[0], tb[1], ray, tmin2)) {                          // :91
                    if (STATS) n_tri++;
                    const float4* tv = reinterpret_cast<const float4*>(&s.triangles[tri]);
                    float u = 0.0f, v = 0.0f;
                    const float dist = ray_triangle(ray, tv[0], tv[1], tv[2], u, v);
                    if (dist < best_t) {                                          // :95
                        best_t = dist; best_tri = tri; best_u = u; best_v = v;
                    }
                }
            }
        }
        if (sp > (uint32_t)kStackDepth) sp = kStackDepth;   // unreachable for unique keys
    }
    return make_float4(best_t, __uint_as_float(best_tri), best_u, best_v);
}

template <bool STATS>
__global__ __launch_bounds__(64) void trace_reference_kernel(trace_args a, lbvh_scene s,
                                                             lbvh_hit* __restrict__ hits,
                                                             lbvh_trace_stats* stats)
{
    __shared__ uint32_t s_stack[kStackDepth][LBVH_WAVE];
    const uint32_t lane = threadIdx.x;
    const uint32_t tile = shard_tile(blockIdx.x, a.shard_index, a.shard_count);
    uint32_t px, py;
    const bool active = tile < a.tiles_x * a.tiles_y && tile_pixel(a, tile, lane, px, py);

    uint32_t n_pops = 0, n_box = 0, n_leaf = 0, n_tri = 0, n_hit = 0;
    if (active) {
        const ray_t ray = make_ray(a.cam, px, py);
        const float4 out = reference_ray<STATS>(s, ray, s_stack, lane, n_pops, n_box, n_leaf, n_tri);
        reinterpret_cast<float4*>(hits)[hit_slot(a, blockIdx.x, lane, px, py)] = out;
        if (STATS && out.x < LBVH_MAX_FLOAT) n_hit = 1;
    }
    if (STATS) add_stats(stats, n_pops, n_box, n_leaf, n_tri, n_hit);
}

// ---------------------------------------------------------------------------------------------
// LBVH_TRACE_FAST_EXACT: which of several triangles hit at exactly the same t the reference keeps
// ---------------------------------------------------------------------------------------------
// The reference keeps the triangle its loop meets FIRST (strict `<`, Raytracing.compute:95).  Its loop (:138-176) tests a node's
// leaf children at once — left, then right — and pushes the internal ones, left then right, so the right subtree is popped
// before the left one.  That is a fixed order of the leaves, whatever the ray (a box that is missed only removes leaves from
// it): at every node, [left child if it is a leaf] [right child if a leaf] [right subtree] [left subtree].  Two leaves are
// ordered by the first node at which their paths from the root part, so a leaf's rank is the string of two-bit digits along its
// path — 0 / 1: the leaf itself as left / right child, 2: into the right subtree, 3: into the left one — compared from the root.
// The reference's own stack holds 64 entries, so a path has at most 64 digits: 128 bits.  Climbing from the leaf, every digit
// is shifted in at the top, so the root's digit ends up most significant.
struct tie_key { unsigned long long hi, lo; };
__device__ __forceinline__ bool key_less(const tie_key& a, const tie_key& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

// (two leaves at once: a climb is a chain of dependent loads — ONE per level: a node's record holds its parent's index — and
// two chains in flight cost what one does)
__device__ __forceinline__ void reference_visit_keys(const lbvh_scene& s, uint32_t leaf_a, uint32_t leaf_b, tie_key& ka, tie_key& kb)
{
    tie_key k[2] = {{0ull, 0ull}, {0ull, 0ull}};
    auto shift_in = [&](int i, unsigned long long digit) { k[i].lo = (k[i].lo >> 2) | (k[i].hi << 62); k[i].hi = (k[i].hi >> 2) | (digit << 62); };
    struct rec3 { uint32_t left, ltype, parent; };
    auto load = [&](uint32_t node) {             // {left, ltype, right, rtype, parent, index}: the words a climb needs
        const uint32_t* nd = reinterpret_cast<const uint32_t*>(&s.internal_nodes[node]);
        const uint2 lt = *reinterpret_cast<const uint2*>(nd);
        return rec3{lt.x, lt.y, nd[4]};
    };
    const uint32_t leaf[2] = {leaf_a, leaf_b};
    uint32_t node[2];
    rec3 rec[2];
#pragma unroll
    for (int i = 0; i < 2; i++) node[i] = reinterpret_cast<const uint32_t*>(&s.leaf_nodes[leaf[i]])[0];           // {parent, index}
#pragma unroll
    for (int i = 0; i < 2; i++) {
        rec[i] = load(node[i]);
        shift_in(i, (rec[i].left == leaf[i] && rec[i].ltype == LBVH_LEAF_NODE) ? 0ull : 1ull);
    }
    for (int guard = 0; guard < 64 && (node[0] != 0u || node[1] != 0u); guard++) {
        rec3 up[2];
#pragma unroll
        for (int i = 0; i < 2; i++) up[i] = load(node[i] != 0u ? rec[i].parent : 0u);
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (node[i] == 0u) continue;
            shift_in(i, (up[i].left == node[i] && up[i].ltype == LBVH_INTERNAL_NODE) ? 3ull : 2ull);
            node[i] = rec[i].parent;
            rec[i] = up[i];
        }
    }
    ka = k[0];
    kb = k[1];
}

constexpr uint32_t kTiePosition = 0x80000000u;       // a record's triangle word still holds a leaf position (resolve pending)

// pass 1: every listed candidate whose t is the record's t tries to take the record (a compare-and-swap on the record's
// triangle word, which holds the leaf position of the candidate in the lead)
__global__ __launch_bounds__(256) void resolve_ties_lead_kernel(trace_args a, lbvh_scene s, lbvh_hit* __restrict__ hits)
{
    const uint32_t count = a.ties[a.tie_turn];
    if (count > a.tie_capacity) return;                      // the list ran over: the second kernel takes every ray that saw a tie
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const uint32_t slot = a.ties[4 + 4 * i], pos = a.ties[6 + 4 * i], tbits = a.ties[7 + 4 * i];
        uint32_t* rec = reinterpret_cast<uint32_t*>(&hits[slot]);
        if (rec[0] != tbits) continue;                       // a tie at a t that was beaten later
        // (system scope: the frame may be another GPU's memory — one frame from N GPUs)
        uint32_t lead = __hip_atomic_load(&rec[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        while (lead != (pos | kTiePosition)) {
            tie_key mine, theirs;
            reference_visit_keys(s, pos, lead & ~kTiePosition, mine, theirs);
            if (!key_less(mine, theirs)) break;
            if (__hip_atomic_compare_exchange_strong(&rec[1], &lead, pos | kTiePosition, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                break;
        }
    }
}

// pass 2 (a launch later: every lead is final): the candidate in the lead writes the record as the reference computes it.
// The list holds one candidate per ray of the launch; a scene that ties more often than that (three and more coincident
// triangles on most pixels) runs it over, and then nothing of it is used: every ray that saw a tie — its record's triangle word
// still carries kTiePosition — goes through the reference's own loop (slow, and only then).  Last of the frame's kernels: it
// clears the NEXT frame's counter.
__global__ __launch_bounds__(64) void resolve_ties_write_kernel(trace_args a, lbvh_scene s, uint32_t n_work, lbvh_hit* __restrict__ hits)
{
    __shared__ uint32_t s_stack[kStackDepth][LBVH_WAVE];
    const uint32_t lane = threadIdx.x;
    const uint32_t count = a.ties[a.tie_turn];
    if (blockIdx.x == 0 && lane == 0) a.ties[a.tie_turn ^ 1u] = 0u;
    if (count <= a.tie_capacity) {
        for (uint32_t i = blockIdx.x * LBVH_WAVE + lane; i < count; i += gridDim.x * LBVH_WAVE) {
            const uint32_t slot = a.ties[4 + 4 * i], xy = a.ties[5 + 4 * i], pos = a.ties[6 + 4 * i], tbits = a.ties[7 + 4 * i];
            uint32_t* rec = reinterpret_cast<uint32_t*>(&hits[slot]);
            if (rec[0] != tbits || rec[1] != (pos | kTiePosition)) continue;
            const uint32_t tri = s.sorted_indices[pos];
            const ray_t ray = make_ray(a.cam, xy & 0xFFFFu, xy >> 16);
            const float4* tv = reinterpret_cast<const float4*>(&s.triangles[tri]);
            float u = 0.0f, v = 0.0f;
            const float dist = ray_triangle(ray, tv[0], tv[1], tv[2], u, v);
            reinterpret_cast<float4*>(hits)[slot] = make_float4(dist, __uint_as_float(tri), u, v);
        }
        return;
    }
    for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const uint32_t tile = shard_tile(w, a.shard_index, a.shard_count);
        uint32_t px, py;
        if (!(tile < a.tiles_x * a.tiles_y && tile_pixel(a, tile, lane, px, py))) continue;
        const size_t slot = hit_slot(a, w, lane, px, py);
        if ((reinterpret_cast<const uint32_t*>(&hits[slot])[1] & kTiePosition) == 0u) continue;
        uint32_t unused = 0;
        reinterpret_cast<float4*>(hits)[slot] = reference_ray<false>(s, make_ray(a.cam, px, py), s_stack, lane, unused, unused, unused, unused);
    }
}

// one candidate per lane of `mask` onto the list (called where the lanes of `mask` are active)
__device__ __forceinline__ void list_tie(const trace_args& a, uint64_t mask, uint32_t w, uint32_t px, uint32_t py, uint32_t pos, float t)
{
    const uint32_t lane = lane_id();
    // (the record slot and the pixel are formed here, in the rare branch: the walk keeps nothing alive for them but the
    // pixel coordinates, which its last store needs anyway)
    const uint32_t slot = (uint32_t)hit_slot(a, w, lane, px, py), xy = px | (py << 16);
    const int first = __builtin_ctzll(mask);
    uint32_t base = 0;
    if ((int)lane == first) base = atomicAdd(&a.ties[a.tie_turn], (uint32_t)__popcll(mask));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
    const uint32_t i = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (i < a.tie_capacity) {
        a.ties[4 + 4 * i] = slot; a.ties[5 + 4 * i] = xy; a.ties[6 + 4 * i] = pos; a.ties[7 + 4 * i] = __float_as_uint(t);
    }
}

// ---------------------------------------------------------------------------------------------
// derived fast-traversal scene
// ---------------------------------------------------------------------------------------------
// lbvh_build_fast_scene's triangle lines (lbvh_build_scene gets them from the Morton kernel instead): one line per
// triangle in the caller's order.  Four lanes per triangle: lane q of a quad reads float4 q of the 128-byte reference
// triangle (a, b, c: the 48 contiguous bytes of its positions) and writes float4 q of the 64-byte line, so a wave's
// one store covers 16 whole lines (1 KB contiguous); a / b / c travel inside the quad by DPP quad_perm.
__global__ __launch_bounds__(256) void build_fast_tris_kernel(const lbvh_triangle* __restrict__ triangles,
                                                              lbvh_fast_tri* __restrict__ tris, uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tri = t >> 2, q = t & 3u;
    const bool live = tri < n;
    float4 mine = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live && q < 3u) mine = reinterpret_cast<const float4*>(&triangles[tri])[q];
    // quad_perm:[k,k,k,k] = 0x00 / 0x55 / 0xAA: every lane of the quad reads lane k's value (all lanes active here)
#define LBVH_QUAD(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
#define LBVH_QUAD3(name, ctrl) \
    const float name##x = LBVH_QUAD(mine.x, ctrl), name##y = LBVH_QUAD(mine.y, ctrl), name##z = LBVH_QUAD(mine.z, ctrl)
    LBVH_QUAD3(a, 0x00);
    LBVH_QUAD3(b, 0x55);
    LBVH_QUAD3(c, 0xAA);
#undef LBVH_QUAD3
#undef LBVH_QUAD
    const float e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    const float e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    // layout: lbvh_common.h — {a, index | a, e2.x | e1, e2.y | e1, e2.z}
    const float4 out = q == 0u ? make_float4(ax, ay, az, __uint_as_float(tri))
                     : q == 1u ? make_float4(ax, ay, az, e2x)
                     : q == 2u ? make_float4(e1x, e1y, e1z, e2y)
                               : make_float4(e1x, e1y, e1z, e2z);
    if (live) reinterpret_cast<float4*>(&tris[tri])[q] = out;
}

// ---------------------------------------------------------------------------------------------
// LBVH_TRACE_FAST — packet traversal
//
// The primary rays of a small pixel tile (8x8 pixels: one ray per lane) leave one pinhole and stay
// together almost to the leaves, so the wave walks the tree ONCE for all of them:
//   * the current node index is wave-uniform: the 64-byte fused node arrives as ONE coalesced line in every
//     16-lane row (lane k loads dword k & 15; address arithmetic on the scalar unit), not as 64 divergent vector
//     gathers (the per-lane form measured 22 % lane utilisation and an L1 pipe stalled on pending misses; more
//     waves per CU made it slower).  The box planes go into the tests as DPP row-broadcast operands straight from
//     that register, only the two child references (and a triangle's values) are moved to SGPRs by v_readlane;
//   * every lane tests ITS ray against the node's two child boxes; a child is entered when any lane
//     hits it (ballot) and its entry t is not beyond that lane's best hit; order = majority vote of
//     the lanes that hit both;
//   * the traversal stack is shared by the wave: one VGPR used as a 64-slot array written by v_writelane and read
//     by v_readlane with a scalar stack pointer — no LDS, no scratch;
//   * all control flow is scalar (conditions come from ballots).
// The kernel is bound by instruction issue (68 vector + 49 scalar instructions per step, vector
// issue 88-91 % busy over the frame), so the step is written for instruction count: see walk_packet's SIGNS form.
// Each lane still sees every node it would visit alone (it votes for it), the leaf's own AABB slab
// test gates the triangle test per lane, and the accept rule is the reference's strict t < best —
// so per-ray results equal the reference order's min t.
// ---------------------------------------------------------------------------------------------
// Work distribution (measured, 1 M triangles / 1080p; per-tile steps at 16x8 pixels: median 48, p99 323, max 594):
//   * persistent waves pulling tiles from per-XCD atomic queues: 1.13 ms — 0.70 ms of it with the traversal
//     switched off (two dependent device-scope round trips per tile on eight hot counters);
//   * one tile per wave, handed out by the hardware dispatcher in row-major order: 0.65 ms;
//   * the same with the tiles dispatched HEAVIEST FIRST: 0.49 ms (lightest first 0.67, random 0.66, screen
//     centre first 0.58): the run time is the tail of late-starting heavy tiles.  The kernel therefore records
//     every tile's step count, and the next trace of the same frame layout dispatches in descending order of
//     those counts (a scheduling hint only — any order gives the same hits; the first frame is row-major):
//     file_tiles_kernel files the tiles under 16 half-octave cost classes (6 us), wave w takes the w-th tile
//     of the class lists, heaviest class first.
//     Cheaper bookkeeping was tried and lost: a "heavy tiles" list built by the trace kernel + row-major for
//     the rest leaves medium tiles in the tail (0.33 instead of 0.27 ms at 8x8 tiles); filing every tile under
//     one of 8 cost classes with a returning atomic per tile: 0.48 ms (hot device-scope counters again).
//   * cutting packets at a step budget and re-queueing their unfinished subtrees as (tile, subtree) items for
//     further passes (results merged per pixel by compare-and-swap) was built and measured: no gain over
//     heaviest-first (0.49 ms either way), so it is not in the code.
// ---------------------------------------------------------------------------------------------
struct uniform_node {        // one fused node, wave-uniform (lives in SGPRs)
    float4 lmin, lmax, rmin, rmax;
};

// Wave-uniform fetch of a 64-byte line WITHOUT the scalar cache (s_load of a ~100 MB working set serialises on
// the scalar cache's miss path, which several CUs share — measured again at the end of round 1: 0.315 vs 0.285 ms):
// lane k loads dword k & 15, i.e. one coalesced 64-byte line per 16-lane row through the CU's vector L1.
#define LBVH_RL(v, k) __int_as_float(__builtin_amdgcn_readlane((v), (k)))
// a 64-byte line (node or triangle) by reference: the address is base + (index << 6), computed on the scalar
// unit; the load takes it as an SGPR base + the lane's constant byte offset (no vector address arithmetic in the walk)
__device__ __forceinline__ int fetch_line_dword(const lbvh_fast_node* __restrict__ lines, uint32_t ref, uint32_t lane_bytes)
{
    const char* base = reinterpret_cast<const char*>(lines) + ((size_t)(ref & 0x7FFFFFFFu) << 6);
    return *reinterpret_cast<const int*>(base + lane_bytes);
}
// the same through a buffer resource over the line array (below 4 GB): the line's byte offset is the instruction's
// scalar offset operand and the lane's constant its vector offset — no address arithmetic on the vector unit at all
struct line_source {
    const lbvh_fast_node* lines;
    __amdgpu_buffer_rsrc_t rsrc;
};
template <bool BUF>
__device__ __forceinline__ int fetch_line(const line_source& src, uint32_t ref, uint32_t lane_bytes)
{
    if (BUF) return (int)__builtin_amdgcn_raw_buffer_load_b32(src.rsrc, (int)lane_bytes, (int)((ref & 0x7FFFFFFFu) << 6), 0);
    return fetch_line_dword(src.lines, ref, lane_bytes);
}

// dword K of the line held by every 16-lane row of w, in all lanes (DPP row_newbcast:K, gfx90a+; folds into the
// consuming vector instruction).  ONLY where every lane is active: a DPP read of a disabled source lane yields 0
// (the triangle test under `if (hit)` tried it and lost the triangle indices)
template <int K>
__device__ __forceinline__ float row_dword(int w)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, w, 0x150 + K, 0xf, 0xf, true));
}
template <int K>
__device__ __forceinline__ float4 row_box(int w)     // dwords K, K + 1, K + 2 as a box corner
{
    return make_float4(row_dword<K>(w), row_dword<K + 1>(w), row_dword<K + 2>(w), 0.0f);
}
__device__ __forceinline__ uniform_node broadcast_node(int w)
{
    uniform_node nd;
    nd.lmin = make_float4(LBVH_RL(w, 0), LBVH_RL(w, 1), LBVH_RL(w, 2), LBVH_RL(w, 3));
    nd.lmax = make_float4(LBVH_RL(w, 4), LBVH_RL(w, 5), LBVH_RL(w, 6), LBVH_RL(w, 7));
    nd.rmin = make_float4(LBVH_RL(w, 8), LBVH_RL(w, 9), LBVH_RL(w, 10), LBVH_RL(w, 11));
    nd.rmax = make_float4(LBVH_RL(w, 12), LBVH_RL(w, 13), LBVH_RL(w, 14), LBVH_RL(w, 15));
    return nd;
}

// same for a triangle line (lbvh_fast_tri: v0 and index in dwords 0-3, e1 in 8-10, e2 in 7, 11, 15)
__device__ __forceinline__ void broadcast_tri(int w, float4& v0, float4& v1, float4& v2)
{
    v0 = make_float4(LBVH_RL(w, 0), LBVH_RL(w, 1), LBVH_RL(w, 2), LBVH_RL(w, 3));
    v1 = make_float4(LBVH_RL(w, 8), LBVH_RL(w, 9), LBVH_RL(w, 10), 0.0f);
    v2 = make_float4(LBVH_RL(w, 7), LBVH_RL(w, 11), LBVH_RL(w, 15), 0.0f);
}

// RX x RY rays per lane: the packet is an (8 RX) x (8 RY)-pixel tile, lane (lx, ly) owns the RX x RY
// pixel block at (lx RX, ly RY).  One tree walk, one node fetch and one vote serve all 64 RX RY rays.
template <int R>
struct packet_rays {
    ray_t ray[R];
    bool act[R];
    float best_t[R], best_u[R], best_v[R];
    uint32_t best_tri[R];
};

struct walk_counters { uint32_t pops, box, leaf, tri; };

// The accept rule of LBVH_TRACE_FAST: the reference's strict `t < best` (Raytracing.compute:95) and, among triangles hit
// at EXACTLY the same t, the lowest triangle index.  The reference keeps whichever of them its own visit order meets
// first; a walk that visits in another order (near child first here, several waves at once on heavy tiles) needs a rule
// that does not depend on the order, or tri / u / v of such a pixel would change with the dispatch history and the
// shard count.  A miss carries t = MAX_FLOAT and index 0, so it never wins a tie.
__device__ __forceinline__ bool closer(float dist, uint32_t tri, float best_t, uint32_t best_tri)
{
    return dist < best_t || (dist == best_t && tri < best_tri);
}

// Walk the tree for one packet; returns the number of steps (node fetches).
// stack[slot] := value, both wave-uniform: one v_writelane_b32 (the lane select goes through M0: a vector instruction
// takes one SGPR operand; both operands come from the scalar unit, so there is no lane-select hazard)
__device__ __forceinline__ void push_slot(int& stack, uint32_t value, uint32_t slot)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(stack) : "s"(value), "s"(slot) : "m0");
#pragma clang diagnostic pop
}

// population count of a lane mask as a 32-bit scalar (two s_bcnt1_i32_b32: the 64-bit form comes back as a 64-bit
// value, and ordered 64-bit compares only exist on the vector unit)
__device__ __forceinline__ int mask_count(uint64_t m)
{
    return __builtin_popcount((uint32_t)m) + __builtin_popcount((uint32_t)(m >> 32));
}

// Do all active rays of the packet agree on the sign of every direction component, with finite non-zero inverse
// directions (every tile but those on the image's centre lines)?  neg: bit k = component k is negative.
__device__ __forceinline__ bool packet_signs(const packet_rays<1>& P, uint32_t& neg)
{
    const ray_t& r = P.ray[0];
    const uint64_t act = __ballot(P.act[0]);
    const float inv[3] = {r.ix, r.iy, r.iz};
    bool ordered = true;
    neg = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint64_t finite = __ballot(P.act[0] && fabsf(inv[k]) < INFINITY && inv[k] != 0.0f);
        const uint64_t negative = __ballot(P.act[0] && inv[k] < 0.0f);
        ordered = ordered && finite == act && (negative == 0 || negative == act);
        neg |= negative != 0 ? 1u << k : 0u;
    }
    return ordered;
}

// which dword of a NODE line a lane fetches: with ordered signs the min and max plane of every axis whose direction
// component is negative change places, so that dwords 0-2 / 8-10 of the line in the register are the planes the
// rays meet first and 4-6 / 12-14 the ones they leave through — the ordering costs nothing per step
__device__ __forceinline__ uint32_t node_line_bytes(uint32_t lane, bool ordered, uint32_t neg)
{
    const uint32_t d = lane & 15u, axis = d & 3u;
    const bool flip = ordered && axis < 3u && ((neg >> axis) & 1u) != 0;
    return (flip ? d ^ 4u : d) * 4u;
}

// EXACT (LBVH_TRACE_FAST_EXACT, one ray per lane): best_tri holds a LEAF POSITION (from the parent's line: dwords 11 / 15), every
// candidate that loses a tie at exactly the best t is listed at once, and `tied` collects the lanes that saw one
struct tie_sink { const trace_args* a; uint32_t w, px, py; uint64_t tied; };

// SIGNS: all active rays of the packet share the sign of each direction component (and have finite non-zero
// inverse directions): bit i of `neg` = component i is negative.  The near / far planes of both child boxes are
// then picked on the scalar unit and the box tests lose their six min / max each (ray_box_ordered).
template <bool STATS, int R, bool SIGNS, bool BUF, bool EXACT = false>
__device__ __forceinline__ uint32_t walk_packet(const line_source& src, packet_rays<R>& P, walk_counters& C, uint32_t neg,
                                                tie_sink* sink = nullptr)
{
    static_assert(!EXACT || R == 1, "the exact mode walks one ray per lane");
    const uint32_t lane = lane_id();
    bool tied = false;
    int stack = 0;            // wave-shared stack: slot k lives in lane k of this VGPR
    uint32_t sp = 0;          // scalar
    uint32_t steps = 0;
    const uint32_t node_bytes = node_line_bytes(lane, SIGNS, neg);      // triangle lines read the same under it
    // root: its own box is never tested, both children are
    int w_node = fetch_line<BUF>(src, 0u, node_bytes);
    for (;;) {
        // SIGNS: node lines arrive with their planes already ordered (node_bytes above): *min = near, *max = far
        const uniform_node nd = broadcast_node(w_node);
        const uint32_t lref = __float_as_uint(nd.lmin.w), rref = __float_as_uint(nd.lmax.w);
        const bool leaf_l = (lref & 0x80000000u) != 0, leaf_r = (rref & 0x80000000u) != 0;
        // both children are fetched NOW (node line or triangle line), before the box tests: whichever
        // the packet goes to next is already in flight — one memory latency per step instead of two
        const int w_l = fetch_line<BUF>(src, lref, node_bytes);
        const int w_r = fetch_line<BUF>(src, rref, node_bytes);
        if (STATS && lane == 0) C.pops++;
        steps++;
        float tl[R], tr[R];
        bool hit_l[R], hit_r[R];
        bool any_l = false, any_r = false;
        int pref = 0;     // > 0: this lane's rays that want both children reach the left one first
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (SIGNS) {
                // the planes come straight out of the line register: every 16-lane row holds the whole line, and a DPP
                // row broadcast feeds dword k to the subtraction as its operand — no v_readlane, no SGPR
                // (evaluated by every lane, inactive ones too: a DPP operand must not sit behind a lane-dependent branch)
                const bool box_l = ray_box_ordered(row_dword<0>(w_node), row_dword<1>(w_node), row_dword<2>(w_node),
                                                   row_dword<4>(w_node), row_dword<5>(w_node), row_dword<6>(w_node), P.ray[r], tl[r]);
                const bool box_r = ray_box_ordered(row_dword<8>(w_node), row_dword<9>(w_node), row_dword<10>(w_node),
                                                   row_dword<12>(w_node), row_dword<13>(w_node), row_dword<14>(w_node), P.ray[r], tr[r]);
                hit_l[r] = P.act[r] & box_l;
                hit_r[r] = P.act[r] & box_r;
            } else {
                const bool box_l = ray_box(row_box<0>(w_node), row_box<4>(w_node), P.ray[r], tl[r]);
                const bool box_r = ray_box(row_box<8>(w_node), row_box<12>(w_node), P.ray[r], tr[r]);
                hit_l[r] = P.act[r] & box_l;
                hit_r[r] = P.act[r] & box_r;
            }
            if (STATS) C.box += (hit_l[r] ? 1u : 0u) + (hit_r[r] ? 1u : 0u);
            // a box that starts beyond this ray's best hit cannot hold a nearer one
            hit_l[r] = hit_l[r] && !(tl[r] > P.best_t[r]);
            hit_r[r] = hit_r[r] && !(tr[r] > P.best_t[r]);
            any_l |= hit_l[r];
            any_r |= hit_r[r];
        }
        // leaves first: their hits tighten best_t before anything is entered
        if (leaf_l && __any(any_l)) {
            float4 v0, v1, v2;
            broadcast_tri(w_l, v0, v1, v2);
            if (STATS && lane == 0) C.leaf++;        // one 48-B triangle fetch for the packet
            any_r = false;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (hit_l[r]) {
                    if (STATS) C.tri++;
                    float u = 0.0f, v = 0.0f;
                    const float dist = ray_fast_triangle(P.ray[r], v0, v1, v2, u, v);
                    const uint32_t id = EXACT ? __float_as_uint(nd.rmin.w) : __float_as_uint(v0.w);      // EXACT: the left leaf's position
                    const bool counts = hit_counts(dist, tl[r]);
                    if (EXACT && counts && dist == P.best_t[r] && dist != LBVH_MAX_FLOAT) {
                        tied = true;                                       // the candidate that does not stay in the running is listed now
                        list_tie(*sink->a, __ballot(true), sink->w, sink->px, sink->py, max(id, P.best_tri[r]), dist);
                    }
                    if (counts && closer(dist, id, P.best_t[r], P.best_tri[r])) { P.best_t[r] = dist; P.best_tri[r] = id; P.best_u[r] = u; P.best_v[r] = v; }
                }
                hit_r[r] = hit_r[r] && !(tr[r] > P.best_t[r]);
                any_r |= hit_r[r];
            }
        }
        if (leaf_r && __any(any_r)) {
            float4 v0, v1, v2;
            broadcast_tri(w_r, v0, v1, v2);
            if (STATS && lane == 0) C.leaf++;
            any_l = false;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (hit_r[r]) {
                    if (STATS) C.tri++;
                    float u = 0.0f, v = 0.0f;
                    const float dist = ray_fast_triangle(P.ray[r], v0, v1, v2, u, v);
                    const uint32_t id = EXACT ? __float_as_uint(nd.rmax.w) : __float_as_uint(v0.w);      // EXACT: the right leaf's position
                    const bool counts = hit_counts(dist, tr[r]);
                    if (EXACT && counts && dist == P.best_t[r] && dist != LBVH_MAX_FLOAT) {
                        tied = true;
                        list_tie(*sink->a, __ballot(true), sink->w, sink->px, sink->py, max(id, P.best_tri[r]), dist);
                    }
                    if (counts && closer(dist, id, P.best_t[r], P.best_tri[r])) { P.best_t[r] = dist; P.best_tri[r] = id; P.best_u[r] = u; P.best_v[r] = v; }
                }
                hit_l[r] = hit_l[r] && !(tl[r] > P.best_t[r]);
                any_l |= hit_l[r];
            }
        }
        if (R != 1) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (hit_l[r] && hit_r[r]) pref += tl[r] <= tr[r] ? 1 : -1;
        }
        const uint64_t ml = leaf_l ? 0ull : __ballot(any_l);
        const uint64_t mr = leaf_r ? 0ull : __ballot(any_r);
        if (ml != 0 && mr != 0) {
            // both children wanted: the side most lanes reach first goes first.  One ray per lane: the votes are
            // population counts of scalar masks (one vector compare, the rest on the scalar unit)
            int l_votes, r_votes;
            if (R == 1) {
                const uint64_t both = ml & mr, le = __ballot(tl[0] <= tr[0]);
                l_votes = mask_count(both & le);
                r_votes = mask_count(both & ~le);
            } else {
                l_votes = __popcll(__ballot(pref > 0));
                r_votes = __popcll(__ballot(pref < 0));
            }
            const int l_lanes = mask_count(ml), r_lanes = mask_count(mr);
            // more votes, or on a tie more lanes (an integer select: a select between two conditions would be
            // materialised in vector registers)
            const int by_votes = l_votes - r_votes, by_lanes = l_lanes - r_lanes;
            const bool l_near = (by_votes != 0 ? by_votes : by_lanes) >= 0;
            const uint32_t far = l_near ? rref : lref;
            w_node = l_near ? w_l : w_r;
            push_slot(stack, far, sp & 63u);            // slot sp := far
            sp++;
        } else if (ml != 0) {
            w_node = w_l;
        } else if (mr != 0) {
            w_node = w_r;
        } else {
            if (sp == 0) {
                if (EXACT) sink->tied = __ballot(tied);
                return steps;
            }
            sp--;
            w_node = fetch_line<BUF>(src, (uint32_t)__builtin_amdgcn_readlane(stack, sp & 63u), node_bytes);
        }
    }
}

// ---- the lean step: primary rays (one origin), ordered signs ---------------------------------------------------------
// Measured instruction costs on gfx950 with >= 4 waves per SIMD (tools/ubench/instcost.hip, profiles/r3): v_mul / v_add /
// v_sub / v_mov / v_fma on VGPRs issue in 2.3 cycles per wave; anything with a DPP or SGPR operand, every v_cmp,
// v_min / v_max / min3 / max3, v_readlane and v_cndmask (SGPR mask) in 4.3; v_cndmask with VCC in 23; one scalar
// instruction per cycle per CU (4 SIMDs share it).  A step of the generic walker above is ~69 vector + ~50 scalar
// instructions: the four SIMDs of a CU need the scalar unit for as long as they need their own vector pipes.  This form
// is written against that table:
//   * all rays of a primary packet leave ONE origin, so `plane - origin` is formed once for the whole line (lane k holds
//     plane k: one full-rate v_sub) and each ray's 12 products take the difference as a DPP operand — 12 v_mul_dpp
//     instead of 12 v_sub_dpp + 12 v_mul, the same two roundings per product ((plane - o) * inv), bit-identical;
//   * `tmax > tmin && tmax > 0` is `tmax > max(tmin, 0)` (no NaNs with finite non-zero inverse directions); lanes
//     without a ray carry best = -inf, which no box entry can precede: the `active` mask is gone from the step;
//   * lane masks stay in SGPRs from the compare to the branch (no bool -> v_cndmask -> v_cmp round trips), the best-hit
//     update is four moves under the accept mask, the uniform select of the next node line names its SGPR mask.
// Same nodes visited, same triangles tested in the same order with the same arithmetic as walk_packet<.., SIGNS = true>.
__device__ __forceinline__ int select_line(int if_set, int if_clear, bool uniform_cond)
{
    // v_cndmask_b32 with an SGPR-pair mask (4.3 cycles); the VCC form the compiler picks for a uniform select costs 23
    const uint64_t m = __builtin_amdgcn_ballot_w64(uniform_cond);
    int out;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(if_clear), "v"(if_set), "s"(m));
    return out;
}

// population count of a lane mask as ONE scalar instruction with a 32-bit result (the builtin's 64-bit result type makes
// the compiler count the halves separately: mask_count above)
__device__ __forceinline__ int lanes_in(uint64_t m)
{
    int out;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(out) : "s"(m) : "scc");
    return out;
}

// max(x, 0) as ONE v_max_f32 (fmaxf's NaN canonicalisation costs a second one; x is never a NaN here)
__device__ __forceinline__ float max_zero(float x)
{
    float out;
    asm("v_max_f32 %0, 0, %1" : "=v"(out) : "v"(x));
    return out;
}

struct lean_tri { float tvx, tvy, tvz, e1x, e1y, e1z, e2x, e2y, e2z; uint32_t index; };

// the triangle line `w` (this lane's dword under the sign-ordered fetch pattern) -> its values in every lane's VECTOR
// registers by DPP row broadcasts (v_mov_b32_dpp, 4.3 cycles each — what a v_readlane costs — but every product that
// uses them afterwards is a VGPR x VGPR instruction at the full rate instead of a half-rate one with an SGPR operand;
// each value is used three times).  All lanes must be active here (a DPP read of a switched-off lane yields 0): the
// caller runs the whole test on every lane and masks the result.  tvec = origin - first vertex comes out of ONE
// subtraction over the line (Raytracing.compute:52: tvec = ray.origin - v0)
__device__ __forceinline__ lean_tri uniform_tri(int w, float o_lane)
{
    const int tv = __float_as_int(o_lane - __int_as_float(w));
    lean_tri t;
    t.tvx = row_dword<0>(tv); t.tvy = row_dword<1>(tv); t.tvz = row_dword<2>(tv);
    t.index = (uint32_t)__builtin_amdgcn_readlane(w, 3);
    t.e1x = row_dword<8>(w); t.e1y = row_dword<9>(w); t.e1z = row_dword<10>(w);
    t.e2x = row_dword<7>(w); t.e2y = row_dword<11>(w); t.e2z = row_dword<15>(w);
    return t;
}

// RayTriangleIntersection (Raytracing.compute:37-73) with tvec given; the operation order of ray_triangle_edges.
// Returns "a hit" and its t / u / v; nothing is selected or written on a miss.
__device__ __forceinline__ bool lean_triangle(const ray_t& r, const lean_tri& T, float& t_out, float& u_out, float& v_out)
{
    const float px = r.dy * T.e2z - r.dz * T.e2y;
    const float py = r.dz * T.e2x - r.dx * T.e2z;
    const float pz = r.dx * T.e2y - r.dy * T.e2x;
    const float det = dot3(T.e1x, T.e1y, T.e1z, px, py, pz);
    const float inv_det = 1.0f / det;
    const float u = dot3(T.tvx, T.tvy, T.tvz, px, py, pz) * inv_det;
    const float qx = T.tvy * T.e1z - T.tvz * T.e1y;
    const float qy = T.tvz * T.e1x - T.tvx * T.e1z;
    const float qz = T.tvx * T.e1y - T.tvy * T.e1x;
    const float v = dot3(r.dx, r.dy, r.dz, qx, qy, qz) * inv_det;
    t_out = dot3(T.e2x, T.e2y, T.e2z, qx, qy, qz) * inv_det;
    u_out = u;
    v_out = v;
    // :49 |det| < 1e-8, :55 u outside [0, 1], :62 v < 0 or u + v > 1 (comparisons false for NaN as in the reference's order)
    return !(det < 1e-8f && det > -1e-8f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f);
}

template <bool BUF, bool EXACT = false>
__device__ __forceinline__ uint32_t walk_packet_lean(const line_source& src, packet_rays<1>& P, uint32_t neg, tie_sink* sink = nullptr)
{
    uint64_t tied = 0;        // EXACT: lanes that met a second triangle at exactly their best t (see walk_packet)
    const uint32_t lane = lane_id();
    const ray_t r = P.ray[0];
    int stack = 0;            // wave-shared stack: slot k lives in lane k of this VGPR
    uint32_t sp = 0, steps = 0;
    const uint32_t node_bytes = node_line_bytes(lane, true, neg);
    const uint32_t axis = lane & 3u;          // dword k of a line is a plane (or a vertex coordinate) of axis k & 3; 3: no plane
    const float o_lane = axis == 0u ? r.ox : (axis == 1u ? r.oy : (axis == 2u ? r.oz : 0.0f));
    float best_t = P.act[0] ? P.best_t[0] : -INFINITY;
    uint32_t best_tri = P.best_tri[0];
    float best_u = P.best_u[0], best_v = P.best_v[0];
    int w_node = fetch_line<BUF>(src, 0u, node_bytes);
    for (;;) {
        const uint32_t lref = (uint32_t)__builtin_amdgcn_readlane(w_node, 3), rref = (uint32_t)__builtin_amdgcn_readlane(w_node, 7);
        const bool leaf_l = (int)lref < 0, leaf_r = (int)rref < 0;
        const int w_l = fetch_line<BUF>(src, lref, node_bytes);        // both children in flight before the tests
        const int w_r = fetch_line<BUF>(src, rref, node_bytes);
        steps++;
        const int d = __float_as_int(__int_as_float(w_node) - o_lane);     // every plane minus the origin, once
        const float tl = fmaxf(row_dword<0>(d) * r.ix, fmaxf(row_dword<1>(d) * r.iy, row_dword<2>(d) * r.iz));
        const float fl = fminf(row_dword<4>(d) * r.ix, fminf(row_dword<5>(d) * r.iy, row_dword<6>(d) * r.iz));
        const float tr = fmaxf(row_dword<8>(d) * r.ix, fmaxf(row_dword<9>(d) * r.iy, row_dword<10>(d) * r.iz));
        const float fr = fminf(row_dword<12>(d) * r.ix, fminf(row_dword<13>(d) * r.iy, row_dword<14>(d) * r.iz));
        // (a mask is the AND of the ballots of single compares: the ballot of an AND goes through v_cndmask + v_cmp)
        const bool box_l = fl > max_zero(tl), box_r = fr > max_zero(tr);
        const bool near_l = !(tl > best_t), near_r = !(tr > best_t);
        bool hit_l = box_l && near_l, hit_r = box_r && near_r;
        uint64_t ml = __builtin_amdgcn_ballot_w64(box_l) & __builtin_amdgcn_ballot_w64(near_l);
        uint64_t mr = __builtin_amdgcn_ballot_w64(box_r) & __builtin_amdgcn_ballot_w64(near_r);
        // leaves first: their hits tighten best_t before anything is entered
        if (leaf_l) {
            if (ml != 0) {
                lean_tri T = uniform_tri(w_l, o_lane);
                if (EXACT) T.index = (uint32_t)__builtin_amdgcn_readlane(w_node, 11);      // the left leaf's position
                float t, u, v;            // every lane computes; the lanes that hit the leaf's box may keep the result
                // (the accept rule's second half, lbvh_rt.h hit_counts: EXACT needs it before the tie ballot; otherwise it sits behind
                // `closer`, under the mask of the lanes that have a candidate at all: skipped with them)
                const bool cand = lean_triangle(r, T, t, u, v) && hit_l && (!EXACT || hit_counts(t, tl));
                if (EXACT) {
                    const uint64_t m = __builtin_amdgcn_ballot_w64(cand) & __builtin_amdgcn_ballot_w64(t == best_t);
                    if (m != 0) {         // (rare) the candidate that drops out of an exact tie is listed now
                        tied |= m;
                        if ((m >> lane) & 1ull) list_tie(*sink->a, m, sink->w, sink->px, sink->py, max(T.index, best_tri), t);
                    }
                }
                if (cand && closer(t, T.index, best_t, best_tri) && (EXACT || hit_counts(t, tl))) { best_t = t; best_tri = T.index; best_u = u; best_v = v; }
                const bool still_r = !(tr > best_t);
                hit_r = hit_r && still_r;
                mr &= __builtin_amdgcn_ballot_w64(still_r);
            }
            ml = 0;
        }
        if (leaf_r) {
            if (mr != 0) {
                lean_tri T = uniform_tri(w_r, o_lane);
                if (EXACT) T.index = (uint32_t)__builtin_amdgcn_readlane(w_node, 15);      // the right leaf's position
                float t, u, v;
                // (the accept rule's second half, lbvh_rt.h hit_counts: EXACT needs it before the tie ballot; otherwise it sits behind
                // `closer`, under the mask of the lanes that have a candidate at all: skipped with them)
                const bool cand = lean_triangle(r, T, t, u, v) && hit_r && (!EXACT || hit_counts(t, tr));
                if (EXACT) {
                    const uint64_t m = __builtin_amdgcn_ballot_w64(cand) & __builtin_amdgcn_ballot_w64(t == best_t);
                    if (m != 0) {
                        tied |= m;
                        if ((m >> lane) & 1ull) list_tie(*sink->a, m, sink->w, sink->px, sink->py, max(T.index, best_tri), t);
                    }
                }
                if (cand && closer(t, T.index, best_t, best_tri) && (EXACT || hit_counts(t, tr))) { best_t = t; best_tri = T.index; best_u = u; best_v = v; }
                ml &= __builtin_amdgcn_ballot_w64(!(tl > best_t));
            }
            mr = 0;
        }
        if (ml != 0 && mr != 0) {
            const uint64_t both = ml & mr, le = __builtin_amdgcn_ballot_w64(tl <= tr);
            const int by_votes = lanes_in(both & le) - lanes_in(both & ~le), by_lanes = lanes_in(ml) - lanes_in(mr);
            const bool l_near = (by_votes != 0 ? by_votes : by_lanes) >= 0;
            push_slot(stack, l_near ? rref : lref, sp & 63u);
            sp++;
            w_node = select_line(w_l, w_r, l_near);
        } else if (ml != 0) {
            w_node = w_l;
        } else if (mr != 0) {
            w_node = w_r;
        } else {
            if (sp == 0) break;
            sp--;
            w_node = fetch_line<BUF>(src, (uint32_t)__builtin_amdgcn_readlane(stack, sp & 63u), node_bytes);
        }
    }
    P.best_t[0] = best_t; P.best_tri[0] = best_tri; P.best_u[0] = best_u; P.best_v[0] = best_v;
    if (EXACT) sink->tied = tied;
    return steps;
}

// do all active rays of the packet leave the same point (primary rays of a pinhole camera do)?
__device__ __forceinline__ bool packet_one_origin(const packet_rays<1>& P)
{
    const ray_t& r = P.ray[0];
    const uint64_t act = __ballot(P.act[0]);
    if (act == 0) return false;
    const int first = __builtin_ctzll(act);
    const float ox = LBVH_RL(__float_as_int(r.ox), first), oy = LBVH_RL(__float_as_int(r.oy), first), oz = LBVH_RL(__float_as_int(r.oz), first);
    return __ballot(P.act[0] && r.ox == ox && r.oy == oy && r.oz == oz) == act;
}

template <int RX, int RY>
__device__ __forceinline__ void tile_rays(const trace_args& a, uint32_t tile, uint32_t lane, packet_rays<RX * RY>& P,
                                          uint32_t& px0, uint32_t& py0)
{
    const uint32_t ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    px0 = (uint32_t)a.x0 + tx * (8u * RX) + (lane & 7u) * RX;
    py0 = (uint32_t)a.y0 + ty * (8u * RY) + (lane >> 3) * RY;
#pragma unroll
    for (int r = 0; r < RX * RY; r++) {
        const uint32_t px = px0 + (uint32_t)(r % RX), py = py0 + (uint32_t)(r / RX);
        P.act[r] = px < (uint32_t)a.x1 && py < (uint32_t)a.y1;
        P.ray[r] = make_ray(a.cam, px, py);
    }
}

// ---- heavy tiles: one WORKGROUP per tile -------------------------------------------------------------------
// With one wave per tile the run time of a frame (or of one GPU's share of it: 212 us at 1/8 of the frame against
// 287 us for all of it) is the critical path of its heaviest tile — hundreds of dependent fetch -> test -> vote
// steps.  Tiles the previous trace filed under the heavy cost classes are therefore walked by kCoopWaves waves
// together: every wave walks its own chain with a private stack, the packet's 64 best hits live in LDS (64-bit
// atomic min on (ordered t, line index of the triangle): pruning is shared by all waves), and a wave with spare
// stack entries hands its OLDEST one (the largest unvisited subtree) to an idle wave through a small LDS list.
constexpr int kCoopWaves = 8;              // waves of a cooperative workgroup in a share of a frame
constexpr int kCoopWavesWhole = 4;         // ... in a whole frame (every wave slot taken: workgroups of 4 waves place like the plain kernel's)
constexpr uint32_t kHeavyClassWhole = 10;   // whole frames: classes >= this (>= 256 steps) are walked cooperatively
constexpr int kHeavyClass = 7;             // cost classes >= this (>= 96 steps) are walked cooperatively
constexpr uint32_t kCoopGrain = 32;        // steps of the last trace per cooperating wave
constexpr uint32_t kCoopMaxWork = 12288;   // tiles per launch up to which every heavy class is walked cooperatively
constexpr uint32_t kSharedMaxWork = 24576; // ... and up to which the very heaviest are (beyond: one wave per tile only)
constexpr uint32_t kHeavyClassFull = 9;    // above that: only classes >= this (>= 192 steps; half frame: 152 us against 184 / 160 / 193 with 8 / 10 / 11)
constexpr uint32_t kLightFrameSteps = 640000; // whole frames whose estimated steps (class lower bounds: an underestimate) stay below this walk their heavy tiles with kCoopWaves
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMovedCoopMaxWork = 6144;  // ... when the camera has moved since the costs were recorded (launch_packets)

struct coop_params { uint32_t cap, first_class, grain; };

__device__ __forceinline__ uint32_t heavy_items(const uint32_t* __restrict__ counts, coop_params cp)
{
    uint32_t h = 0;
#pragma unroll
    for (int c = 0; c < kOrderClasses; c++) h += (uint32_t)c >= cp.first_class ? counts[c] : 0u;
    return min(h, cp.cap);
}

// floats <-> unsigned keys with the same order (t may be negative: the reference has no t > 0 test)
__device__ __forceinline__ uint32_t ordered_key(float t)
{
    const uint32_t b = __float_as_uint(t);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k)
{
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

struct coop_shared {
    unsigned long long best[64];     // per ray: ordered t << 32 | line index of the triangle
    uint32_t give[32];               // subtrees on offer
    uint32_t give_n, lock, idle, steps;
    unsigned long long ties;         // EXACT: rays that met two triangles at exactly the same t
};

__device__ __forceinline__ void coop_lock(coop_shared& S)
{
    while (atomicCAS(&S.lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void coop_unlock(coop_shared& S)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(&S.lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one heavy tile (work item w) walked by the waves of this workgroup
// EXACT (LBVH_TRACE_FAST_EXACT): the shared keys carry leaf POSITIONS (from the parent's line) instead of line indices, the
// candidate that loses an exact tie in the atomic minimum is listed at once, and the tile's writer leaves the rays that saw a tie
// to the resolve kernels (see walk_packet / light_tile)
template <bool STATS, int WAVES, bool EXACT = false>
__device__ __forceinline__ void coop_tile(coop_shared& S, const trace_args& a, const lbvh_fast_node* __restrict__ nodes,
                                          const lbvh_fast_tri* __restrict__ tris, uint32_t n_work, uint32_t w,
                                          coop_params heavy_cap, uint32_t* __restrict__ cost, lbvh_hit* __restrict__ hits,
                                          lbvh_trace_stats* stats, uint32_t* __restrict__ tile_cost)
{
    const uint32_t tile = shard_tile(w, a.shard_index, a.shard_count);
    if (w >= n_work || tile >= a.tiles_x * a.tiles_y) return;        // cannot happen for a heavy item
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    // as many waves as the tile's last step count is worth (about kCoopGrain steps each); the others leave now
    const uint32_t n_waves = min(max((cost[w] + heavy_cap.grain / 2u) / heavy_cap.grain, 2u), (uint32_t)WAVES);
    if (wave >= n_waves) return;
    packet_rays<1> P;
    uint32_t px0, py0;
    tile_rays<1, 1>(a, tile, lane, P, px0, py0);
    uint32_t neg = 0;
    const bool ordered = packet_signs(P, neg);      // see walk_packet
    const uint32_t node_bytes = node_line_bytes(lane, ordered, neg);
    if (threadIdx.x < 64u) S.best[threadIdx.x] = (unsigned long long)ordered_key(LBVH_MAX_FLOAT) << 32;
    if (threadIdx.x == 0) { S.give_n = 0; S.lock = 0; S.idle = n_waves - 1u; S.steps = 0; S.ties = 0; }
    __syncthreads();

    volatile coop_shared& V = S;
    int stack = 0;                  // private stack: entries [base, sp) live in lanes (index & 63) of this VGPR
    uint32_t base = 0, sp = 0;
    bool have = wave == 0;          // wave 0 starts at the root, the others wait for an offer
    uint32_t cur = 0;
    walk_counters C = {0, 0, 0, 0};
    uint32_t steps = 0;
    bool tied = false;
    for (;;) {
        if (!have) {
            uint32_t got = kNone, all_idle = 0;
            if (lane == 0) {
                if (V.give_n > 0) {
                    coop_lock(S);
                    if (V.give_n > 0) {
                        got = V.give[V.give_n - 1];
                        V.give_n = V.give_n - 1;
                        atomicSub(&S.idle, 1u);          // idle is also bumped outside the lock (atomically)
                    }
                    coop_unlock(S);
                } else {
                    all_idle = V.idle == n_waves ? 1u : 0u;
                }
            }
            got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            all_idle = (uint32_t)__builtin_amdgcn_readfirstlane((int)all_idle);
            if (got == kNone) {
                if (all_idle) break;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
            cur = got;
            have = true;
        }
        int w_node = fetch_line_dword(nodes, cur, node_bytes);
        for (;;) {      // one chain: until this wave has nothing left
            const uniform_node nd = broadcast_node(w_node);
            const uint32_t lref = __float_as_uint(nd.lmin.w), rref = __float_as_uint(nd.lmax.w);
            const bool leaf_l = (lref & 0x80000000u) != 0, leaf_r = (rref & 0x80000000u) != 0;
            const int w_l = fetch_line_dword(nodes, lref, node_bytes);
            const int w_r = fetch_line_dword(nodes, rref, node_bytes);
            if (STATS && lane == 0) C.pops++;
            steps++;
            float best_t = key_value((uint32_t)(V.best[lane] >> 32));      // everybody's hits so far
            float tl, tr;
            bool hit_l, hit_r;
            if (ordered) {
                hit_l = ray_box_ordered(row_dword<0>(w_node), row_dword<1>(w_node), row_dword<2>(w_node),
                                        row_dword<4>(w_node), row_dword<5>(w_node), row_dword<6>(w_node), P.ray[0], tl);
                hit_r = ray_box_ordered(row_dword<8>(w_node), row_dword<9>(w_node), row_dword<10>(w_node),
                                        row_dword<12>(w_node), row_dword<13>(w_node), row_dword<14>(w_node), P.ray[0], tr);
            } else {
                hit_l = ray_box(row_box<0>(w_node), row_box<4>(w_node), P.ray[0], tl);
                hit_r = ray_box(row_box<8>(w_node), row_box<12>(w_node), P.ray[0], tr);
            }
            hit_l = P.act[0] & hit_l;
            hit_r = P.act[0] & hit_r;
            if (STATS) C.box += (hit_l ? 1u : 0u) + (hit_r ? 1u : 0u);
            hit_l = hit_l && !(tl > best_t);
            hit_r = hit_r && !(tr > best_t);
            if (leaf_l && __any(hit_l)) {
                float4 v0, v1, v2;
                broadcast_tri(w_l, v0, v1, v2);
                if (STATS && lane == 0) C.leaf++;
                if (hit_l) {
                    if (STATS) C.tri++;
                    float u = 0.0f, v = 0.0f;
                    const float dist = ray_fast_triangle(P.ray[0], v0, v1, v2, u, v);
                    if (dist <= best_t && dist != LBVH_MAX_FLOAT && hit_counts(dist, tl)) {    // ties go to the atomic: (t, line index) orders them (see closer())
                        best_t = dist;
                        const uint32_t id = EXACT ? (uint32_t)__builtin_amdgcn_readlane(w_node, 11) : (lref & 0x7FFFFFFFu);
                        const unsigned long long key = ((unsigned long long)ordered_key(dist) << 32) | id;
                        const unsigned long long was = atomicMin(&S.best[lane], key);
                        if (EXACT && (uint32_t)(was >> 32) == (uint32_t)(key >> 32) && was != key) {
                            tied = true;                           // the candidate that is not kept joins the list
                            list_tie(a, __ballot(true), w, px0, py0, (uint32_t)max(was, key), dist);
                        }
                    }
                }
                hit_r = hit_r && !(tr > best_t);
            }
            if (leaf_r && __any(hit_r)) {
                float4 v0, v1, v2;
                broadcast_tri(w_r, v0, v1, v2);
                if (STATS && lane == 0) C.leaf++;
                if (hit_r) {
                    if (STATS) C.tri++;
                    float u = 0.0f, v = 0.0f;
                    const float dist = ray_fast_triangle(P.ray[0], v0, v1, v2, u, v);
                    if (dist <= best_t && dist != LBVH_MAX_FLOAT && hit_counts(dist, tr)) {    // ties go to the atomic: (t, line index) orders them (see closer())
                        best_t = dist;
                        const uint32_t id = EXACT ? (uint32_t)__builtin_amdgcn_readlane(w_node, 15) : (rref & 0x7FFFFFFFu);
                        const unsigned long long key = ((unsigned long long)ordered_key(dist) << 32) | id;
                        const unsigned long long was = atomicMin(&S.best[lane], key);
                        if (EXACT && (uint32_t)(was >> 32) == (uint32_t)(key >> 32) && was != key) {
                            tied = true;
                            list_tie(a, __ballot(true), w, px0, py0, (uint32_t)max(was, key), dist);
                        }
                    }
                }
                hit_l = hit_l && !(tl > best_t);
            }
            const uint64_t ml = leaf_l ? 0ull : __ballot(hit_l);
            const uint64_t mr = leaf_r ? 0ull : __ballot(hit_r);
            bool more = true;
            if (ml != 0 && mr != 0) {
                const uint64_t both = ml & mr, le = __ballot(tl <= tr);       // votes as in walk_packet
                const int by_votes = mask_count(both & le) - mask_count(both & ~le), by_lanes = mask_count(ml) - mask_count(mr);
                const bool l_near = (by_votes != 0 ? by_votes : by_lanes) >= 0;
                const uint32_t far = l_near ? rref : lref;
                w_node = l_near ? w_l : w_r;
                // (this walker's stack pointer lives in a vector register: hand the asm scalar copies)
                push_slot(stack, (uint32_t)__builtin_amdgcn_readfirstlane((int)far),
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(sp & 63u)));
                sp++;
            } else if (ml != 0) {
                w_node = w_l;
            } else if (mr != 0) {
                w_node = w_r;
            } else if (sp != base) {
                sp--;
                w_node = fetch_line_dword(nodes, (uint32_t)__builtin_amdgcn_readlane(stack, sp & 63u), node_bytes);
            } else {
                more = false;
            }
            // an idle wave and a spare entry: offer my oldest one (the largest subtree I still hold)
            if (sp != base) {
                uint32_t offer = 0;
                if (lane == 0) offer = (V.idle > V.give_n && V.give_n < 32u) ? 1u : 0u;
                if (__builtin_amdgcn_readfirstlane((int)offer)) {
                    const uint32_t node = (uint32_t)__builtin_amdgcn_readlane(stack, base & 63u);
                    uint32_t given = 0;
                    if (lane == 0) {
                        coop_lock(S);
                        if (V.give_n < 32u) { V.give[V.give_n] = node; V.give_n = V.give_n + 1; given = 1; }
                        coop_unlock(S);
                    }
                    base += (uint32_t)__builtin_amdgcn_readfirstlane((int)given);
                }
            }
            if (!more) break;
        }
        have = false;
        if (lane == 0) atomicAdd(&S.idle, 1u);
    }
    const uint64_t tied_lanes = EXACT ? __ballot(tied) : 0ull;
    if (lane == 0) {
        atomicAdd(&S.steps, steps);
        if (EXACT && tied_lanes) atomicOr(&S.ties, (unsigned long long)tied_lanes);
    }
    __syncthreads();
    uint32_t n_hit = 0;
    if (wave == 0) {
        if (lane == 0) {
            cost[w] = S.steps;
            if (STATS && tile_cost) tile_cost[tile] = S.steps;
        }
        if (P.act[0]) {
            const unsigned long long key = S.best[lane];
            const float t = key_value((uint32_t)(key >> 32));
            float4 out = make_float4(LBVH_MAX_FLOAT, __uint_as_float(0u), 0.0f, 0.0f);
            if (t < LBVH_MAX_FLOAT) {
                // barycentrics (and the original index) from the winning triangle: the same arithmetic as in the walk
                const lbvh_fast_node* line = EXACT ? reinterpret_cast<const lbvh_fast_node*>(tris) + a.sorted_indices[(uint32_t)key]
                                                   : &nodes[(uint32_t)key];                                // a triangle line
                float4 v0, v1, v2;
                unpack_fast_triangle(reinterpret_cast<const float4*>(line), v0, v1, v2);
                float u = 0.0f, v = 0.0f;
                const float dist = ray_fast_triangle(P.ray[0], v0, v1, v2, u, v);
                out = make_float4(dist, v0.w, u, v);
                if (STATS) n_hit++;
            }
            if (EXACT) {
                // a ray that saw a tie: its candidate in the lead joins the list and stays in the record as a position
                const uint64_t tm = S.ties;
                const bool mine = P.act[0] && ((tm >> lane) & 1ull) && t < LBVH_MAX_FLOAT;
                const uint64_t mm = __ballot(mine);
                if (mine) {
                    list_tie(a, mm, w, px0, py0, (uint32_t)key, t);
                    out.y = __uint_as_float((uint32_t)key | kTiePosition);
                }
            }
            reinterpret_cast<float4*>(hits)[hit_slot(a, w, lane, px0, py0)] = out;
        }
    }
    if (STATS) add_stats(stats, C.pops, C.box, C.leaf, C.tri, n_hit);
}

// one tile (work item w) walked by one wave
template <bool STATS, bool EXACT = false>
__device__ __forceinline__ void light_tile(const trace_args& a, const lbvh_fast_node* __restrict__ nodes,
                                           const lbvh_fast_tri* __restrict__ tris, uint32_t w, uint32_t lane,
                                           uint32_t* __restrict__ cost, lbvh_hit* __restrict__ hits, lbvh_trace_stats* stats,
                                           uint32_t* __restrict__ tile_cost)
{
    const uint32_t tile = shard_tile(w, a.shard_index, a.shard_count);
    if (tile >= a.tiles_x * a.tiles_y) {                         // tail of the last group
        if (lane == 0) cost[w] = 0;
        return;
    }
    packet_rays<1> P;
    uint32_t px0, py0;
    tile_rays<1, 1>(a, tile, lane, P, px0, py0);
    P.best_t[0] = LBVH_MAX_FLOAT; P.best_tri[0] = 0; P.best_u[0] = 0.0f; P.best_v[0] = 0.0f;
    walk_counters C = {0, 0, 0, 0};
    uint32_t neg = 0;
    const bool ordered = packet_signs(P, neg);
    line_source src;
    src.lines = nodes;
    src.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<lbvh_fast_node*>(nodes), 0, (int)a.line_bytes, 0x00020000);
    uint32_t steps;
    tie_sink sink = {&a, 0u, 0u, 0u, 0ull};
    if (EXACT) {
        // the walkers with the tie bookkeeping (closer() on leaf positions: the loser of every exact tie is listed)
        sink.w = w; sink.px = px0; sink.py = py0;
        if (ordered && packet_one_origin(P))
            steps = a.line_bytes != 0 ? walk_packet_lean<true, true>(src, P, neg, &sink) : walk_packet_lean<false, true>(src, P, neg, &sink);
        else if (a.line_bytes != 0)
            steps = ordered ? walk_packet<false, 1, true, true, true>(src, P, C, neg, &sink) : walk_packet<false, 1, false, true, true>(src, P, C, 0u, &sink);
        else
            steps = ordered ? walk_packet<false, 1, true, false, true>(src, P, C, neg, &sink) : walk_packet<false, 1, false, false, true>(src, P, C, 0u, &sink);
        // a ray that saw a tie: its candidate in the lead joins the list and stays in the record as a position until the
        // resolve kernels have run; every other ray's position becomes its triangle here
        const bool mine = (sink.tied >> lane) & 1ull;
        if (mine) list_tie(a, sink.tied, w, px0, py0, P.best_tri[0], P.best_t[0]);
        if (P.act[0] && P.best_t[0] < LBVH_MAX_FLOAT) P.best_tri[0] = mine ? (P.best_tri[0] | kTiePosition) : a.sorted_indices[P.best_tri[0]];
    } else
    if (!STATS && ordered && packet_one_origin(P))
        steps = a.line_bytes != 0 ? walk_packet_lean<true>(src, P, neg) : walk_packet_lean<false>(src, P, neg);
    else if (a.line_bytes != 0)
        steps = ordered ? walk_packet<STATS, 1, true, true>(src, P, C, neg) : walk_packet<STATS, 1, false, true>(src, P, C, 0u);
    else
        steps = ordered ? walk_packet<STATS, 1, true, false>(src, P, C, neg) : walk_packet<STATS, 1, false, false>(src, P, C, 0u);
    if (lane == 0) cost[w] = steps;
    if (STATS && tile_cost && lane == 0) tile_cost[tile] = steps;
    uint32_t n_hit = 0;
    if (P.act[0]) {
        float4 out;
        out.x = P.best_t[0];
        out.y = __uint_as_float(P.best_tri[0]);
        out.z = P.best_u[0];
        out.w = P.best_v[0];
        reinterpret_cast<float4*>(hits)[hit_slot(a, w, lane, px0, py0)] = out;
        if (STATS && P.best_t[0] < LBVH_MAX_FLOAT) n_hit++;
    }
    if (STATS) add_stats(stats, C.pops, C.box, C.leaf, C.tri, n_hit);
}

// A full chip (a whole frame on one GPU): one wave per tile, 4 tiles per workgroup, the w-th tile of the class
// lists, heaviest class first.  No cooperative tiles: with every wave slot taken they do not pay (287 -> 304 us).
template <bool STATS, bool EXACT = false>
__global__ __launch_bounds__(256) void trace_packet_kernel(trace_args a, const lbvh_fast_node* __restrict__ nodes,
                                                           const lbvh_fast_tri* __restrict__ tris, uint32_t n_work,
                                                           const uint32_t* __restrict__ counts, const uint32_t* __restrict__ lists,
                                                           uint32_t* __restrict__ cost, lbvh_hit* __restrict__ hits,
                                                           lbvh_trace_stats* stats, uint32_t* __restrict__ tile_cost)
{
    const uint32_t lane = lane_id();
    uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (w >= n_work) return;
    if (counts) {        // the w-th item of the class lists, heaviest class first (they partition the work items)
        uint32_t k = w;
        int c = kOrderClasses - 1;
        for (; c > 0 && k >= counts[c]; c--) k -= counts[c];
        w = lists[(size_t)c * n_work + k];
    } else if (a.shard_count == 1u && a.centre_first && w < a.tiles_x * a.tiles_y) {      // (items past the last tile: the last group's tail)
        // No history (a first frame, another layout): rows and columns are taken centre-out instead of top-down — what costs
        // most is usually in the middle of the picture, and a frame is as long as its heaviest tiles started late
        // (cfg2, first frames: 274 - 284 -> 259 - 261 us on the same box).  A hint like the history's: any order gives the same records.
        auto centre_out = [](uint32_t k, uint32_t n) {
            const uint32_t mid = n / 2u;
            if (n & 1u) return (k & 1u) ? mid + (k + 1u) / 2u : mid - k / 2u;
            return (k & 1u) ? mid + (k - 1u) / 2u : mid - 1u - k / 2u;
        };
        const uint32_t ty = w / a.tiles_x, tx = w - ty * a.tiles_x;
        w = centre_out(ty, a.tiles_y) * a.tiles_x + centre_out(tx, a.tiles_x);
    } else if (a.centre_first && a.shard_count > 1u) {
        // a share of a frame: its groups of 8 tiles (they lie in row-major order over the picture) from the middle one outwards
        auto centre_out = [](uint32_t k, uint32_t n) {
            const uint32_t mid = n / 2u;
            if (n & 1u) return (k & 1u) ? mid + (k + 1u) / 2u : mid - k / 2u;
            return (k & 1u) ? mid + (k - 1u) / 2u : mid - 1u - k / 2u;
        };
        const uint32_t groups = n_work / kShardGroup;          // (n_work is a whole number of groups)
        w = centre_out(w / kShardGroup, groups) * kShardGroup + w % kShardGroup;
    }
    light_tile<STATS, EXACT>(a, nodes, tris, w, lane, cost, hits, stats, tile_cost);
}

// One launch per frame share.  Workgroups of 8 waves; the hardware dispatcher hands them out in index order:
//   * workgroups [0, cap): the heavy tiles of the previous trace, one per workgroup, walked cooperatively
//     (workgroups beyond the actual number of heavy tiles leave at once) — at the FRONT of the grid, so they start
//     first (as a second kernel on another stream they were starved by the light tiles' workgroups);
//   * the rest: one tile per wave, the w-th tile of the class lists after the heavy ones, heaviest class first.
template <bool STATS, int WAVES, bool EXACT = false>
__global__ __launch_bounds__(WAVES * 64) void trace_shared_kernel(trace_args a, const lbvh_fast_node* __restrict__ nodes,
                                                                       const lbvh_fast_tri* __restrict__ tris, uint32_t n_work,
                                                                       const uint32_t* __restrict__ counts,
                                                                       const uint32_t* __restrict__ lists, coop_params heavy_cap,
                                                                       uint32_t* __restrict__ cost, lbvh_hit* __restrict__ hits,
                                                                       lbvh_trace_stats* stats, uint32_t* __restrict__ tile_cost)
{
    __shared__ coop_shared S;
    const uint32_t lane = lane_id();
    const uint32_t heavy = counts ? heavy_items(counts, heavy_cap) : 0u;
    uint32_t w;
    if (blockIdx.x < heavy_cap.cap) {
        if (blockIdx.x >= heavy) return;                         // uniform for the workgroup
        w = blockIdx.x;
    } else {
        w = (blockIdx.x - heavy_cap.cap) * (uint32_t)WAVES + (threadIdx.x >> 6) + heavy;
        if (w >= n_work) return;
    }
    if (counts) {        // the w-th item of the class lists, heaviest class first (they partition the work items)
        uint32_t k = w;
        int c = kOrderClasses - 1;
        for (; c > 0 && k >= counts[c]; c--) k -= counts[c];
        w = lists[(size_t)c * n_work + k];
    }
    if (blockIdx.x < heavy_cap.cap) {
        coop_tile<STATS, WAVES, EXACT>(S, a, nodes, tris, n_work, w, heavy_cap, cost, hits, stats, tile_cost);
        return;
    }
    if (w >= n_work) return;
    light_tile<STATS, EXACT>(a, nodes, tris, w, lane, cost, hits, stats, tile_cost);
}

// Files every work item of the last trace under one of 16 cost classes (half-octave scale): lists[c][...] with
// counts[c].  The next trace's wave w takes the w-th item of the concatenation, heaviest class first.  One
// workgroup per 1024 items; a wave counts / ranks its 64 items per class with a ballot, a workgroup reserves its
// run in each class list with ONE global atomic per class (per-item atomics on 16 hot counters measured 5x the
// cost of the whole traversal's bookkeeping).  The order inside a class is arbitrary: it is a dispatch hint.
__device__ __forceinline__ uint32_t order_class(uint32_t steps)
{
    // 0: < 12, then [12,16) [16,24) [24,32) [32,48) ... : two classes per octave, 15: >= 1536
    if (steps < 12u) return 0u;
    const uint32_t e = 31u - (uint32_t)__builtin_clz(steps);          // >= 3
    const uint32_t half = (steps >> (e - 1u)) & 1u;
    const uint32_t c = 2u * (e - 3u) + half;                          // 12..15 -> 1, 16..23 -> 2, 24..31 -> 3, ...
    return c > 15u ? 15u : c;
}

// The camera has moved since the costs were recorded (`spread` > 0): the costs are looked up where the picture WAS.
// The point on the ray through a tile's centre is taken back into the old camera's frame (q = old world-to-camera x
// new camera-to-world, rotation part; shift = the new position seen from the old camera) and projected with the old
// projection: exact for a turn of the camera, at the image's edge as well as in its centre (1 degree of yaw at 1080p
// is 2.3 tiles in the middle and 4.8 at the left and right edge).  What a change of POSITION does depends on depth,
// which nobody knows here: it is applied at the distance of the scene box's centre, and a tile is filed under the
// largest cost within `spread` tiles of the place it came from (a heavy tile that starts late is the whole frame's
// tail).  Parts of the picture that were outside the old frame take the
// cost of the nearest old tile.  Sharded launches only see their own tiles: places owned by other shards are skipped.
struct tile_grid {
    uint32_t tiles_x, tiles_y, shard_index, shard_count, spread;
    uint32_t reproject;          // 0: look the cost up in place
    float q[9];                  // new camera space -> old camera space (row-major)
    float shift[3];              // new camera position in old camera space
    float depth;                 // distance at which the change of position is applied (the scene box's centre)
    float cam_w, cam_h, near;    // camera-space extent of the image plane (make_ray)
    float px_w, px_h;            // screen_width, screen_height
    int32_t x0, y0;              // the traced rectangle's origin
};

__device__ __forceinline__ uint32_t item_of_tile(const tile_grid& g, uint32_t tile)      // inverse of shard_tile; ~0u: not owned
{
    const uint32_t grp = tile / kShardGroup;
    if (grp % g.shard_count != g.shard_index) return 0xFFFFFFFFu;
    return (grp / g.shard_count) * kShardGroup + tile % kShardGroup;
}

// `frame` (nullable): the step counts of EVERY tile of the previous frame, whoever traced it (lbvh_trace_costs_import):
// then a tile's cost is looked up there instead of among this shard's own items only
__global__ __launch_bounds__(1024) void file_tiles_kernel(const uint32_t* __restrict__ cost, uint32_t n_work,
                                                          uint32_t* __restrict__ counts, uint32_t* __restrict__ lists,
                                                          uint32_t* __restrict__ next_counts, tile_grid grid,
                                                          const uint32_t* __restrict__ frame, uint32_t* work_estimate)
{
    __shared__ uint32_t s_count[kOrderClasses], s_base[kOrderClasses];
    const uint32_t t = threadIdx.x, lane = lane_id();
    const uint32_t i = blockIdx.x * 1024u + t;
    // the class counters take turns: this launch clears the set the NEXT filing will count into (the trace that
    // read it has finished), so no fill kernel sits between the rebuild and the trace.  On the way out that set tells how much
    // work the frame before last was: tiles per class x the class's lower bound, left in a mapped host word for launch_packets
    // (a hint two frames stale: which cooperative workgroup shape a whole frame takes)
    if (blockIdx.x == 0 && t < (uint32_t)kOrderClasses) {
        const uint32_t tiles_in_class = next_counts[t];
        // lower bounds of the classes of order_class: < 12, then 12, 16, 24, 32, 48, ... (two per octave)
        const uint32_t low = t == 0u ? 4u : ((t & 1u) ? 12u << ((t - 1u) / 2u) : 16u << (t / 2u - 1u));
        uint32_t steps = tiles_in_class * low;
#pragma unroll
        for (int d = 1; d < kOrderClasses; d <<= 1) steps += (uint32_t)__shfl_xor((int)steps, d, kOrderClasses);
        if (t == 0 && work_estimate) __hip_atomic_store(work_estimate, steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        next_counts[t] = 0;
    }
    if (t < (uint32_t)kOrderClasses) s_count[t] = 0;
    __syncthreads();
    uint32_t cls = 0xFFFFFFFFu;
    if (i < n_work) {
        uint32_t c = cost[i];
        const uint32_t tile = shard_tile(i, grid.shard_index, grid.shard_count);
        if (grid.spread != 0 && tile < grid.tiles_x * grid.tiles_y) {
            int ty = (int)(tile / grid.tiles_x), tx = (int)(tile - (uint32_t)ty * grid.tiles_x);
            const int r = (int)grid.spread;
            if (grid.reproject) {
                // the ray through the tile's centre (make_ray's camera-space direction), seen from the old camera
                const float px = (float)grid.x0 + (float)tx * 8.0f + 4.0f, py = (float)grid.y0 + (float)ty * 8.0f + 4.0f;
                const float d0 = -0.5f * grid.cam_w + grid.cam_w / grid.px_w * px;
                const float d1 = -0.5f * grid.cam_h + grid.cam_h / grid.px_h * py;
                const float d2 = -grid.near;
                // ... as far out as the middle of the scene: that is where the change of position is charged
                const float s = grid.depth / sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
                const float ox = (grid.q[0] * d0 + grid.q[1] * d1 + grid.q[2] * d2) * s + grid.shift[0];
                const float oy = (grid.q[3] * d0 + grid.q[4] * d1 + grid.q[5] * d2) * s + grid.shift[1];
                const float oz = (grid.q[6] * d0 + grid.q[7] * d1 + grid.q[8] * d2) * s + grid.shift[2];
                if (oz < -1e-6f * grid.near) {                       // in front of the old camera
                    const float k = grid.near / -oz;
                    const float opx = (ox * k + 0.5f * grid.cam_w) * (grid.px_w / grid.cam_w);
                    const float opy = (oy * k + 0.5f * grid.cam_h) * (grid.px_h / grid.cam_h);
                    const float ftx = floorf((opx - (float)grid.x0) * 0.125f), fty = floorf((opy - (float)grid.y0) * 0.125f);
                    tx = (int)fminf(fmaxf(ftx, 0.0f), (float)(grid.tiles_x - 1));
                    ty = (int)fminf(fmaxf(fty, 0.0f), (float)(grid.tiles_y - 1));
                    const uint32_t from = (uint32_t)ty * grid.tiles_x + (uint32_t)tx;
                    if (frame) {
                        c = frame[from];
                    } else {
                        const uint32_t k0 = item_of_tile(grid, from);
                        if (k0 < n_work) c = cost[k0];
                    }
                }
            }
            for (int dy = -r; dy <= r; dy++)
                for (int dx = -r; dx <= r; dx++) {
                    const int x = tx + dx, y = ty + dy;
                    if (x < 0 || y < 0 || x >= (int)grid.tiles_x || y >= (int)grid.tiles_y) continue;
                    const uint32_t at = (uint32_t)y * grid.tiles_x + (uint32_t)x;
                    if (frame) {
                        c = max(c, frame[at]);
                    } else {
                        const uint32_t k = item_of_tile(grid, at);
                        if (k < n_work) c = max(c, cost[k]);
                    }
                }
        }
        cls = order_class(c);
    }
    uint64_t mine = 0;                                  // lanes of this wave in my class
    uint32_t wave_n = 0;                                // lane c < 16: this wave's items of class c
#pragma unroll
    for (int c = 0; c < kOrderClasses; c++) {
        const uint64_t m = __ballot(cls == (uint32_t)c);
        if (cls == (uint32_t)c) mine = m;
        if (lane == (uint32_t)c) wave_n = (uint32_t)__popcll(m);
    }
    uint32_t wave_ofs = 0;                              // lane c: this wave's offset inside the workgroup's run of class c
    if (lane < (uint32_t)kOrderClasses && wave_n) wave_ofs = atomicAdd(&s_count[lane], wave_n);
    __syncthreads();
    if (t < (uint32_t)kOrderClasses && s_count[t])
        s_base[t] = __hip_atomic_fetch_add(&counts[t], s_count[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // wave_ofs of lane `cls` — read with EVERY lane active: ds_bpermute returns 0 for a disabled source lane, and in the
    // ragged last wave the lanes holding the offsets of the higher classes have no item of their own
    const uint32_t ofs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((cls & 15u) << 2), (int)wave_ofs);
    if (cls != 0xFFFFFFFFu) lists[(size_t)cls * n_work + s_base[cls] + ofs + mbcnt64(mine)] = i;
}

// this shard's costs into a full-frame array (tile order), other shards' tiles untouched
__global__ __launch_bounds__(256) void export_costs_kernel(const uint32_t* __restrict__ cost, uint32_t n_work, uint32_t n_tiles,
                                                           uint32_t shard_index, uint32_t shard_count, uint32_t* __restrict__ frame)
{
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= n_work) return;
    const uint32_t tile = shard_tile(w, shard_index, shard_count);
    if (tile < n_tiles) frame[tile] = cost[w];
}

// packed shares -> their pixels of the full frame (the owner's side of lbvh_trace_primary_shard_packed): one wave per
// (share, work item), 16 bytes per lane both ways
__global__ __launch_bounds__(256) void frame_unpack_kernel(const lbvh_hit* __restrict__ packed, uint64_t share_stride, uint32_t first_shard,
                                                           uint32_t shard_count, uint32_t items_per_share, uint32_t tiles_x, uint32_t n_tiles,
                                                           uint32_t width, uint32_t height, lbvh_hit* __restrict__ frame)
{
    const uint32_t lane = lane_id();
    const uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6), s = blockIdx.y;
    if (k >= items_per_share) return;
    const uint32_t tile = shard_tile(k, first_shard + s, shard_count);
    if (tile >= n_tiles) return;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const uint32_t px = tx * 8u + (lane & 7u), py = ty * 8u + (lane >> 3);
    if (px >= width || py >= height) return;
    reinterpret_cast<float4*>(frame)[(size_t)py * width + px] =
        reinterpret_cast<const float4*>(packed)[(size_t)s * share_stride + (size_t)k * 64u + lane];
}

int launch_packets(lbvh_context* ctx, trace_args a, lbvh_hit* d_hits, lbvh_trace_stats* d_stats, uint32_t* d_tile_cost)
{
    a.tiles_x = (uint32_t)(a.x1 - a.x0 + 7) / 8;
    a.tiles_y = (uint32_t)(a.y1 - a.y0 + 7) / 8;
    {
        const uint64_t bytes = (uint64_t)ctx->fast_capacity * (sizeof(lbvh_fast_node) + sizeof(lbvh_fast_tri));
        a.line_bytes = bytes < 0xFFFFFFFFull ? (uint32_t)bytes : 0u;
    }
    const uint32_t n_work = shard_work(a.tiles_x * a.tiles_y, a.shard_index, a.shard_count);
    if (n_work == 0) return LBVH_OK;
    a.centre_first = ctx->debug_switch[LBVH_DEBUG_COLD_ORDER] ? 0u : 1u;      // (measurement switch: include/lbvh_debug.h)
    // [class counts | cost of each work item in the last trace | 16 class lists]: valid for one frame layout
    const size_t cost_bytes = (((size_t)n_work * 4) + 255) & ~(size_t)255;
    void* before = ctx->trace_queues;
    int rc = lbvh_reserve(ctx, &ctx->trace_queues, &ctx->trace_queues_bytes, 256 + cost_bytes + (size_t)kOrderClasses * cost_bytes);
    if (rc != LBVH_OK) return rc;
    if (ctx->trace_queues != before) {
        ctx->trace_history = false;
        ctx->trace_counts_turn = 0;
        LBVH_HIP_TRY(ctx, hipMemsetAsync(ctx->trace_queues, 0, 256, ctx->cur_stream));
    }
    // two sets of 16 class counters in the first 256 bytes
    uint32_t* counts = (uint32_t*)ctx->trace_queues + 32u * ctx->trace_counts_turn;
    uint32_t* next_counts = (uint32_t*)ctx->trace_queues + 32u * (ctx->trace_counts_turn ^ 1u);
    uint32_t* cost = (uint32_t*)((char*)ctx->trace_queues + 256);
    uint32_t* lists = (uint32_t*)((char*)ctx->trace_queues + 256 + cost_bytes);
    const uint64_t layout = ((uint64_t)a.tiles_x << 48) ^ ((uint64_t)a.tiles_y << 32) ^ ((uint64_t)a.shard_index << 16) ^
                            (uint64_t)a.shard_count ^ ((uint64_t)(uint32_t)a.x0 << 8) ^ ((uint64_t)(uint32_t)a.y0 << 24);
    const bool have_history = ctx->trace_layout == layout && ctx->trace_layout_work == n_work && ctx->trace_history;
    // Has the picture moved since the costs were recorded?  Then they are looked up where each tile came from (the
    // rotation between the two cameras, exact) and spread over one tile around that place (rounding to whole tiles,
    // and what a change of position does to near geometry).  Measured at 1080p, yaw 1 degree per frame: 0.315 ms with
    // the stale order as it is, 0.255 with the costs spread over 2 tiles in place (round 2's first form), see DESIGN 7.
    uint32_t spread = 0;
    tile_grid grid = {};
    if (memcmp(&ctx->trace_camera, &a.cam, sizeof(lbvh_camera)) != 0) {
        spread = 1;
        const lbvh_camera& c0 = ctx->trace_camera;
        const lbvh_camera& c1 = a.cam;
        const bool same_lens = c0.screen_width == c1.screen_width && c0.screen_height == c1.screen_height &&
                               c0.camera_fov == c1.camera_fov && c0.near_plane == c1.near_plane;
        // q = inverse(R0) x R1 over the rotation parts (general 3x3 inverse: Unity's matrix carries a z flip)
        const float* m = c0.camera_to_world;
        const double r0[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
        const double det = r0[0] * (r0[4] * r0[8] - r0[5] * r0[7]) - r0[1] * (r0[3] * r0[8] - r0[5] * r0[6]) +
                           r0[2] * (r0[3] * r0[7] - r0[4] * r0[6]);
        if (same_lens && fabs(det) > 1e-12 && c1.camera_fov > 0.0f && c1.near_plane > 0.0f) {
            const double inv[9] = {(r0[4] * r0[8] - r0[5] * r0[7]) / det, (r0[2] * r0[7] - r0[1] * r0[8]) / det, (r0[1] * r0[5] - r0[2] * r0[4]) / det,
                                   (r0[5] * r0[6] - r0[3] * r0[8]) / det, (r0[0] * r0[8] - r0[2] * r0[6]) / det, (r0[2] * r0[3] - r0[0] * r0[5]) / det,
                                   (r0[3] * r0[7] - r0[4] * r0[6]) / det, (r0[1] * r0[6] - r0[0] * r0[7]) / det, (r0[0] * r0[4] - r0[1] * r0[3]) / det};
            const float* n = c1.camera_to_world;
            const double r1[9] = {n[0], n[1], n[2], n[4], n[5], n[6], n[8], n[9], n[10]};
            bool finite = true;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    const double v = inv[3 * r + 0] * r1[0 + c] + inv[3 * r + 1] * r1[3 + c] + inv[3 * r + 2] * r1[6 + c];
                    grid.q[3 * r + c] = (float)v;
                    finite = finite && std::isfinite(grid.q[3 * r + c]);
                }
            // the new camera's position seen from the old one, and how far out the scene is
            const double dp[3] = {(double)n[3] - m[3], (double)n[7] - m[7], (double)n[11] - m[11]};
            for (int r = 0; r < 3; r++) {
                grid.shift[r] = (float)(inv[3 * r + 0] * dp[0] + inv[3 * r + 1] * dp[1] + inv[3 * r + 2] * dp[2]);
                finite = finite && std::isfinite(grid.shift[r]);
            }
            {
                const double cx = (double)ctx->fast_centre[0] - n[3], cy = (double)ctx->fast_centre[1] - n[7], cz = (double)ctx->fast_centre[2] - n[11];
                grid.depth = (float)sqrt(cx * cx + cy * cy + cz * cz);
                finite = finite && std::isfinite(grid.depth) && grid.depth > 0.0f;
            }
            if (finite) {
                grid.reproject = 1;
                grid.near = c1.near_plane;
                grid.cam_h = 2.0f * c1.near_plane * c1.camera_fov;
                grid.cam_w = (float)c1.screen_width * grid.cam_h / (float)c1.screen_height;
                grid.px_w = (float)c1.screen_width;
                grid.px_h = (float)c1.screen_height;
                grid.x0 = a.x0;
                grid.y0 = a.y0;
            }
        }
    }
    ctx->trace_camera = a.cam;
    // (re-measured after the walk's instruction diet: 1/8 frame 90 us with class 7 against 94 / 109 with 8 / 9;
    // 1/4 frame 139 us with 7 against 117 / 121 with 8 / 9)
    const uint32_t first_class = n_work <= kCoopMaxWork / 2u ? (uint32_t)kHeavyClass
                                 : n_work <= kCoopMaxWork   ? (uint32_t)kHeavyClass + 1u
                                                            : kHeavyClassFull;
    if (have_history) {
        grid.tiles_x = a.tiles_x; grid.tiles_y = a.tiles_y; grid.shard_index = a.shard_index; grid.shard_count = a.shard_count;
        grid.spread = spread;
        // the whole previous frame's costs, if the caller has merged the ranks' (multi-GPU frames under a moving camera)
        const uint32_t* frame = ctx->trace_frame_valid && ctx->trace_frame_tiles_x == a.tiles_x && ctx->trace_frame_tiles_y == a.tiles_y &&
                                        spread != 0 ? ctx->trace_frame_costs : nullptr;
        LBVH_LAUNCH(ctx, file_tiles_kernel, dim3((n_work + 1023) / 1024), dim3(1024), cost, n_work, counts, lists, next_counts, grid, frame,
                    ctx->fault_dev + 24);
        ctx->trace_counts_turn ^= 1u;
    }
    // an imported cost map is a hint for the ONE frame that follows it, used or not (an unused map — static camera,
    // another layout — used to stay valid and could steer a frame traced much later)
    ctx->trace_frame_valid = false;
    // Which tiles are walked cooperatively (known from the last trace).  It costs ~50 % more steps on those tiles (a
    // subtree handed to another wave is walked before the near hits that would have pruned it are known), so it is
    // for under-filled launches — one GPU's share of a multi-GPU frame: every tile of 96 steps or more up to
    // 6 144 tiles, of 128 or more up to 12 288, only those of 192 steps or more up to 24 576, none beyond — a whole
    // 1080p frame runs the plain kernel: 225 us against 234 / 225 with the heaviest classes cooperative (1080p
    // whole / half / quarter / eighth of the frame at the time: 287 -> 275, 234 -> 190, 226 -> 170, 212 -> 103 us).
    // With a camera that has moved the costs are spread over the neighbouring tiles, which is right for the ORDER but
    // makes every neighbour of a heavy tile a cooperative one (+50 % steps each): half a 1080p frame with a 1 degree
    // yaw per frame took 0.34 ms that way against 0.156 static and 0.245 with no history at all.  Such a launch keeps
    // the spread order and walks one wave per tile (0.209 ms); only below 6 144 tiles do cooperative tiles still win
    // (1/2, 1/3, 1/4, 1/8 of the frame, yaw 1 degree: 0.209 / 0.197 / 0.187 / 0.174 ms with this rule, 0.361 / 0.275 /
    // 0.232 / 0.174 with cooperative tiles throughout, 0.245 / 0.196 / 0.233 / 0.190 cold).
    const bool stale_coop = spread != 0 && n_work > kMovedCoopMaxWork;
    // Whole frames (round 3): with the lean step the frame is its heaviest packet's chain, so the classes
    // from 256 steps up (187 of 32 400 tiles at cfg2) are walked cooperatively there too — by workgroups of FOUR waves:
    // the launch's other workgroups (one tile per wave) then place exactly like the plain kernel's, which the 8-wave form
    // does not (its light tiles alone cost 8 %: 0.232 against 0.214 ms).  0.213 -> 0.181 ms (classes >= 11: 0.19 - 0.21).
    // (a moved camera: the plain kernel — the widened costs would make every neighbour of a heavy tile cooperative, 0.27 - 0.29 ms;
    // marking by the un-widened reprojected cost finds too few of them: 0.20 - 0.22 against 0.21 plain)
    // (round 5: with the cooperative set capped at 1/64 of the tiles, cooperative whole frames under a MOVED camera measure the
    // same as the plain kernel — yaw 1 degree 0.207 / 0.208 ms, strafe 0.208 / 0.197 — and with a larger set (1/32, 1/16, 1/8 of
    // the tiles, enough to hold every neighbour of a heavy tile that the widened costs mark) worse: 0.215 / 0.238 / 0.272 ms.
    // The penalty of a moving camera is the prediction of WHICH tiles are heavy, not the lack of cooperation)
    const bool whole = n_work > kSharedMaxWork && spread == 0;
    const bool exact = a.ties != nullptr;          // LBVH_TRACE_FAST_EXACT: the same three launch shapes, kernels with the tie bookkeeping
    if (have_history && whole) {
        // at most 1/64 of the frame's tiles (the heaviest: the lists are heaviest first) — cfg2 has 190 tiles of 256 steps or
        // more among 32 400, but on a scene whose EVERY tile is that heavy (cfg4: 16 M triangles, sub-pixel) a quarter of the frame
        // walked cooperatively cost +50 % steps on all of it: 1.43 ms against 1.01 for the plain kernel (profiles/r5/d_*)
        #ifndef LBVH_WHOLE_COOP_DIV
#define LBVH_WHOLE_COOP_DIV 64u
#endif
        coop_params hp = {std::max(64u, n_work / LBVH_WHOLE_COOP_DIV), kHeavyClassWhole, kCoopGrain};
        // A LIGHT whole frame (the scene far away: few tiles hold all of it, each thousands of steps) leaves most of the chip idle
        // behind its heaviest tiles' chains: there the workgroups of 8 waves that shares of a frame use are the better shape —
        // cfg2 from z = 800 (448 k steps, 2 % of the rays hit): 265 -> 200 us; from z = 400 (815 k): 178 -> 172; with every wave
        // slot taken (z = 250, 160, 0: 1.6 - 2.2 M steps) they cost 5 - 8 %; 16 waves: 248 us at z = 800.  The frame before last's work estimate (mapped host
        // word, left by file_tiles_kernel) decides; nothing known: the 4-wave shape.
        const uint32_t work_estimate = *(const volatile uint32_t*)(ctx->fault_host + 24);
        const bool light = work_estimate < kLightFrameSteps;
        const uint32_t blocks = hp.cap + (light ? (n_work + kCoopWaves - 1) / kCoopWaves : (n_work + kCoopWavesWhole - 1) / kCoopWavesWhole);
        if (light) {
            if (exact)
                LBVH_LAUNCH(ctx, (trace_shared_kernel<false, kCoopWaves, true>), dim3(blocks), dim3(kCoopWaves * 64), a, ctx->fast_nodes, ctx->fast_tris,
                            n_work, counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
            else if (d_stats)
                LBVH_LAUNCH(ctx, (trace_shared_kernel<true, kCoopWaves>), dim3(blocks), dim3(kCoopWaves * 64), a, ctx->fast_nodes, ctx->fast_tris, n_work,
                            counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
            else
                LBVH_LAUNCH(ctx, (trace_shared_kernel<false, kCoopWaves>), dim3(blocks), dim3(kCoopWaves * 64), a, ctx->fast_nodes, ctx->fast_tris, n_work,
                            counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
        } else if (exact)
            LBVH_LAUNCH(ctx, (trace_shared_kernel<false, kCoopWavesWhole, true>), dim3(blocks), dim3(kCoopWavesWhole * 64), a, ctx->fast_nodes,
                        ctx->fast_tris, n_work, counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
        else if (d_stats)
            LBVH_LAUNCH(ctx, (trace_shared_kernel<true, kCoopWavesWhole>), dim3(blocks), dim3(kCoopWavesWhole * 64), a, ctx->fast_nodes, ctx->fast_tris,
                        n_work, counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
        else
            LBVH_LAUNCH(ctx, (trace_shared_kernel<false, kCoopWavesWhole>), dim3(blocks), dim3(kCoopWavesWhole * 64), a, ctx->fast_nodes, ctx->fast_tris,
                        n_work, counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
    } else if (have_history && n_work <= kSharedMaxWork && !stale_coop) {
        coop_params hp = {n_work / 4u, first_class, kCoopGrain};
        const uint32_t blocks = hp.cap + (n_work + kCoopWaves - 1) / kCoopWaves;
        if (exact)
            LBVH_LAUNCH(ctx, (trace_shared_kernel<false, kCoopWaves, true>), dim3(blocks), dim3(kCoopWaves * 64), a, ctx->fast_nodes, ctx->fast_tris,
                        n_work, counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
        else if (d_stats)
            LBVH_LAUNCH(ctx, (trace_shared_kernel<true, kCoopWaves>), dim3(blocks), dim3(kCoopWaves * 64), a, ctx->fast_nodes, ctx->fast_tris, n_work,
                        counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
        else
            LBVH_LAUNCH(ctx, (trace_shared_kernel<false, kCoopWaves>), dim3(blocks), dim3(kCoopWaves * 64), a, ctx->fast_nodes, ctx->fast_tris, n_work,
                        counts, lists, hp, cost, d_hits, d_stats, d_tile_cost);
    } else {
        const uint32_t blocks = (n_work + 3) / 4;
        if (exact)
            LBVH_LAUNCH(ctx, (trace_packet_kernel<false, true>), dim3(blocks), dim3(256), a, ctx->fast_nodes, ctx->fast_tris, n_work,
                        have_history ? counts : nullptr, lists, cost, d_hits, d_stats, d_tile_cost);
        else if (d_stats)
            LBVH_LAUNCH(ctx, trace_packet_kernel<true>, dim3(blocks), dim3(256), a, ctx->fast_nodes, ctx->fast_tris, n_work,
                        have_history ? counts : nullptr, lists, cost, d_hits, d_stats, d_tile_cost);
        else
            LBVH_LAUNCH(ctx, trace_packet_kernel<false>, dim3(blocks), dim3(256), a, ctx->fast_nodes, ctx->fast_tris, n_work,
                        have_history ? counts : nullptr, lists, cost, d_hits, d_stats, d_tile_cost);
    }
    ctx->trace_layout = layout;
    ctx->trace_layout_work = n_work;
    ctx->trace_shard_index = a.shard_index;
    ctx->trace_shard_count = a.shard_count;
    ctx->trace_tiles_x = a.tiles_x;
    ctx->trace_tiles_y = a.tiles_y;
    ctx->trace_origin_x = a.x0;
    ctx->trace_origin_y = a.y0;
    ctx->trace_history = true;
    return LBVH_OK;
}

}  // namespace

// one array of 64-byte lines for the derived scene: [capacity traversal nodes | capacity triangle lines]
static lbvh_status ensure_fast_lines(lbvh_context* ctx, uint32_t n)
{
    if (ctx->fast_capacity >= n) return LBVH_OK;
    LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side_stream) LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->side_stream));
    if (ctx->fast_nodes) { LBVH_HIP_TRY(ctx, hipFree(ctx->fast_nodes)); ctx->fast_nodes = nullptr; }
    ctx->fast_tris = nullptr;
    ctx->fast_capacity = 0;
    ctx->fast_valid = false;
    LBVH_HIP_TRY(ctx, hipMalloc((void**)&ctx->fast_nodes, (size_t)n * (sizeof(lbvh_fast_node) + sizeof(lbvh_fast_tri))));
    ctx->fast_tris = reinterpret_cast<lbvh_fast_tri*>(ctx->fast_nodes + n);
    ctx->fast_capacity = n;
    return LBVH_OK;
}

// parts: 1 = traversal tree + fused nodes, 2 = triangle lines (independent of the tree), 3 = both
static lbvh_status build_fast_scene_parts(lbvh_context* ctx, const lbvh_scene* h_scene, const float h_box_min[3],
                                          const float h_box_max[3], int parts)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, h_scene != nullptr && h_box_min != nullptr && h_box_max != nullptr);
    const lbvh_scene s = *h_scene;
    LBVH_REQUIRE(ctx, s.n >= 2 && s.n <= 0x3FFFFFFFu);     // nodes and triangles share one 31-bit line index
    LBVH_REQUIRE(ctx, s.sorted_indices && s.triangle_aabb && s.internal_nodes && s.leaf_nodes && s.bvh &&
                          s.triangles);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        const lbvh_status arc = ensure_fast_lines(ctx, s.n);
        if (arc != LBVH_OK) return arc;
    }
    // The traversal tree: same sorted triangle order as the scene, its own topology over aligned keys
    // (lbvh_build.hip) and its own boxes.  The scene's internalNodes / leafNodes / bvhData — the
    // reference's bit-exact arrays — are not read here: any tree over the same leaves gives the same
    // hits (every leaf keeps its own AABB test), a tighter one just gives them sooner.
    for (int k = 0; k < 3; k++) ctx->fast_centre[k] = 0.5f * (h_box_min[k] + h_box_max[k]);      // (launch_packets: depth proxy)
    int rc = lbvh_reserve(ctx, &ctx->fast_tree, &ctx->fast_tree_bytes, (size_t)s.n * 4);
    if (rc != LBVH_OK) return rc;
    uint32_t* t_keys = (uint32_t*)ctx->fast_tree;
    if (parts & 1) {
        // the one random gather: triangle AABBs into sorted (leaf) order + the range hierarchy over them + aligned keys
        rc = lbvh_launch_gather_hier(ctx, s.n, s.triangle_aabb, s.sorted_indices, h_box_min, h_box_max, t_keys, true);
        if (rc != LBVH_OK) return rc;
        // lbvh_build_scene: the reference lane's tree kernel takes its boxes from the same hierarchy
        // (the single-workgroup top-levels kernel on lane 0 instead, beside the aligned-keys scan, measured slower: the
        // two extra cross-stream dependencies cost more than the 7 us they take off this lane — 0.293 vs 0.287 ms)
        if (ctx->lane == 1 && ctx->ev_hier) LBVH_HIP_TRY(ctx, hipEventRecord(ctx->ev_hier, ctx->cur_stream));
        // topology + both child boxes of every node in one kernel: the 64-byte traversal nodes
        if ((rc = lbvh_launch_tree_fused(ctx, s.n, t_keys, s.sorted_indices, ctx->fast_nodes, ctx->fast_capacity)) != LBVH_OK) return rc;
    }
    // part 2: the triangle lines (independent of the tree and of the sort)
    if (parts & 2)
        LBVH_LAUNCH(ctx, build_fast_tris_kernel, dim3((s.n + 63) / 64), dim3(256), s.triangles, ctx->fast_tris, s.n);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

extern "C" {

lbvh_status lbvh_build_fast_scene(lbvh_context* ctx, const lbvh_scene* h_scene, const float h_box_min[3],
                                  const float h_box_max[3])
{
    if (ctx) ctx->fast_valid = false;        // the cache is being rewritten: valid again only if all of it is enqueued
    const lbvh_status rc = build_fast_scene_parts(ctx, h_scene, h_box_min, h_box_max, 3);
    if (rc == LBVH_OK) lbvh_note_fast_built(ctx, *h_scene);
    return rc;
}

}  // extern "C"

// RaytracingMeshDrawer.Awake()'s whole build chain: after the sort two independent strands — as three merged launches on one
// stream up to 2 M triangles (lbvh_build.hip "merged launches"), as two concurrent lanes (streams) beyond.
// morton_done: the Morton kernel (keys, indices, triangle AABBs, triangle lines, the sort's cleared scratch) has been enqueued by
// the caller already — lbvh_animate_build_scene's fused animate + Morton kernel, whose arguments change every frame and
// therefore stay outside the replayed graph
static lbvh_status build_scene_enqueue(lbvh_context* ctx, const lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                                       const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                                       lbvh_aabb* d_aabb, lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh,
                                       uint32_t flags, bool morton_done = false)
{
    {
        const int src = lbvh_ensure_side(ctx);
        if (src != LBVH_OK) return src;
    }
    lbvh_status rc;
    // LBVH_BUILD_RESET_NODES (NullLeaf / uint.MaxValue fills, Sc/MeshBufferContainer.cs:114-115): the tree kernel rewrites every word
    // of the nodes below n, so only the slots past the tree and the root's parent word are refilled — by the Morton kernel, on
    // its way (two fills of 24 + 8 MB stood in front of the chain: 13 us of the rebuild)
    const bool reset = (flags & LBVH_BUILD_RESET_NODES) != 0;
    // (argument checks of the public stage functions are repeated here only where lbvh_build_scene's own do not cover them)
    LBVH_REQUIRE(ctx, ((uintptr_t)d_triangles & 15) == 0 && ((uintptr_t)d_aabb & 15) == 0);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_internal & 7) == 0 && ((uintptr_t)d_leaf & 7) == 0 && ((uintptr_t)d_bvh & 15) == 0);
    LBVH_REQUIRE(ctx, n <= 0x7FFFFFFFu && capacity <= 0x3FFFFFFFu);
    {
        // the Morton kernel clears the sort's counters and look-back words, the tree kernels the refit counters:
        // three fill kernels less in the chain
        uint32_t* zero = nullptr;
        uint32_t zero_words = 0;
        if ((rc = (lbvh_status)lbvh_sort_scratch(ctx, capacity, &zero, &zero_words)) != LBVH_OK) return rc;
        lbvh_fast_tri* lines = nullptr;
        if (flags & LBVH_BUILD_FAST_SCENE) {        // the derived scene's triangle lines ride on the Morton kernel
            if ((rc = ensure_fast_lines(ctx, n)) != LBVH_OK) return rc;
            lines = ctx->fast_tris;
        }
        if (!morton_done)
            lbvh_launch_morton(ctx, d_triangles, n, capacity, h_box_min, h_box_max, d_keys, d_indices, d_aabb, zero, zero_words, lines,
                               reset ? d_internal : nullptr, reset ? d_leaf : nullptr);
        // (Morton codes are below 2^30: MeshBufferContainer.cs:41-50; the 0xFFFFFFFF pads are not, and sort last all the same)
        if ((rc = (lbvh_status)lbvh_launch_sort(ctx, d_keys, d_indices, capacity, zero != nullptr, 30u)) != LBVH_OK) return rc;
    }
    const bool fast = (flags & LBVH_BUILD_FAST_SCENE) != 0;
    const bool env_two_streams = ctx->debug_switch[LBVH_DEBUG_BUILD_FORM] >= 3;     // measurement switch: the round 2 - 3 form
    if (fast && !env_two_streams) {
        // both strands of the rest of the chain — derived scene, reference arrays — in three merged launches on this stream
        // (lbvh_build.hip "merged launches"): scenes up to 2 M triangles
        for (int k = 0; k < 3; k++) ctx->fast_centre[k] = 0.5f * (h_box_min[k] + h_box_max[k]);      // as build_fast_scene_parts
        if ((rc = (lbvh_status)lbvh_reserve(ctx, &ctx->fast_tree, &ctx->fast_tree_bytes, (size_t)n * 4)) != LBVH_OK) return rc;
        bool done = false;
        if ((rc = (lbvh_status)lbvh_launch_post_sort_merged(ctx, n, d_keys, d_aabb, d_indices, h_box_min, h_box_max,
                                                            (uint32_t*)ctx->fast_tree, d_internal, d_leaf, d_bvh, ctx->fast_nodes,
                                                            ctx->fast_capacity, &done)) != LBVH_OK)
            return rc;
        if (done) {
            LBVH_HIP_TRY(ctx, hipGetLastError());
            return LBVH_OK;
        }
    }
    if (fast) {
        // lane 1: the derived traversal scene needs only the sorted indices and the triangle AABBs; its first kernels also
        // build the range hierarchy both tree kernels take their boxes from
        LBVH_HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        LBVH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
        lbvh_scene s;
        s.n = n;
        s.sorted_indices = d_indices;
        s.triangle_aabb = d_aabb;
        s.internal_nodes = d_internal;
        s.leaf_nodes = d_leaf;
        s.bvh = d_bvh;
        s.triangles = d_triangles;
        ctx->lane = 1;
        ctx->cur_stream = ctx->side_stream;
        rc = build_fast_scene_parts(ctx, &s, h_box_min, h_box_max, 1);
        hipError_t e = hipEventRecord(ctx->ev_join, ctx->side_stream);
        ctx->lane = 0;
        ctx->cur_stream = ctx->stream;
        if (rc != LBVH_OK) return rc;
        LBVH_HIP_TRY(ctx, e);
    }
    // lane 0: the reference's arrays: DistributeKeys beside lane 1's gather, then TreeConstructor + BVHData in one kernel
    // as soon as the hierarchy is there
    if ((rc = lbvh_distribute_keys(ctx, d_keys, n)) != LBVH_OK) return rc;
    if (fast) {
        LBVH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_hier, 0));
    } else {
        if ((rc = (lbvh_status)lbvh_launch_gather_hier(ctx, n, d_aabb, d_indices, nullptr, nullptr, nullptr, true)) != LBVH_OK) return rc;
    }
    if ((rc = (lbvh_status)lbvh_launch_tree_boxes(ctx, n, d_keys, d_internal, d_leaf, d_bvh)) != LBVH_OK) return rc;
    LBVH_HIP_TRY(ctx, hipGetLastError());
    if (fast) LBVH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return LBVH_OK;
}

extern "C" {

}  // extern "C"

// lbvh_build_scene, and lbvh_animate_build_scene (anim != nullptr): the same chain behind a fused animate + Morton kernel
static lbvh_status build_scene_impl(lbvh_context* ctx, const lbvh_anim* anim, const lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                                    const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                                    lbvh_aabb* d_aabb, lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh,
                                    uint32_t flags)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, n >= 2 && capacity >= n);
    LBVH_REQUIRE(ctx, h_box_min != nullptr && h_box_max != nullptr);
    LBVH_REQUIRE(ctx, d_triangles && d_keys && d_indices && d_aabb && d_internal && d_leaf && d_bvh);
    LBVH_REQUIRE(ctx, ctx->lane == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // FNV-1a over every argument the enqueued work depends on
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { for (int i = 0; i < 8; i++) { key ^= (v >> (8 * i)) & 0xFFu; key *= 1099511628211ull; } };
    mix((uint64_t)(uintptr_t)d_triangles); mix(n); mix(capacity); mix((uint64_t)(uintptr_t)d_keys); mix((uint64_t)(uintptr_t)d_indices);
    mix((uint64_t)(uintptr_t)d_aabb); mix((uint64_t)(uintptr_t)d_internal); mix((uint64_t)(uintptr_t)d_leaf);
    mix((uint64_t)(uintptr_t)d_bvh); mix(flags); mix(anim ? 1u : 0u);
    for (int k = 0; k < 3; k++) { uint32_t b; memcpy(&b, &h_box_min[k], 4); mix(b); memcpy(&b, &h_box_max[k], 4); mix(b); }
    auto mix_scratch = [&](decltype(mix)& m) {      // every context-owned buffer the captured kernels point into
        m((uint64_t)(uintptr_t)ctx->fast_nodes); m((uint64_t)(uintptr_t)ctx->fast_tris); m((uint64_t)(uintptr_t)ctx->fast_tree);
        m((uint64_t)(uintptr_t)ctx->hier);
        m((uint64_t)(uintptr_t)ctx->sort_scratch);
        for (int l = 0; l < 2; l++) { m((uint64_t)(uintptr_t)ctx->scan_scratch[l]); m((uint64_t)(uintptr_t)ctx->refit_scratch[l]); }
    };
    const uint64_t args_key = key;
    mix_scratch(mix);
    if (key == 0) key = 1;
    // debugging / measurement switches (lbvh_debug_switch LBVH_DEBUG_BUILD_FORM: 1 plain always, 2 graph always, 3 two streams plain, 4 two streams as a graph)
    const uint32_t form = ctx->debug_switch[LBVH_DEBUG_BUILD_FORM];
    const bool env_no_graph = form == 1 || form == 3, env_graph = form == 2 || form == 4, env_two_streams = form >= 3;
    // The replayed graph pays where the chain forks onto a second stream (the derived scene beyond the merged launches' size).  On
    // ONE stream — the merged chain's nine launches, the reference arrays alone — kernels enqueued one by one run 5 - 8 us sooner
    // than the graph (rebuild 0.241 against 0.249 ms, reference arrays alone 0.184 against 0.190, step 0.420 against 0.428:
    // profiles/r4/f); the host's launches are hidden behind the GPU's work.
    const bool single_stream = (flags & LBVH_BUILD_FAST_SCENE) == 0 || (lbvh_post_sort_merges(n) && !env_two_streams);
    const bool graphs = ctx->own_stream && !ctx->prof_enabled && !ctx->build_graph_off && !env_no_graph && (!single_stream || env_graph);
    // Host-side state of the call, the same on every path below (plain launches, capture, replay): Morton and the sort
    // rewrite keys / indices / triangle AABBs — whatever derived scene existed is stale — and with
    // LBVH_BUILD_FAST_SCENE the derived scene describes THIS scene once the work is enqueued.
    lbvh_scene built = {};
    built.n = n; built.sorted_indices = d_indices; built.triangle_aabb = d_aabb; built.internal_nodes = d_internal;
    built.leaf_nodes = d_leaf; built.bvh = d_bvh; built.triangles = d_triangles;
    const bool fast_flag = (flags & LBVH_BUILD_FAST_SCENE) != 0;
    lbvh_note_write(ctx, d_indices, (size_t)capacity * 4);
    lbvh_note_write(ctx, d_aabb, (size_t)n * sizeof(lbvh_aabb));
    if (fast_flag) ctx->fast_valid = false;
    if (anim) {
        // the fused animate + Morton kernel: a plain launch in front of the (replayed) rest of the chain — its angle changes
        // every frame.  The sort's scratch it clears and the triangle lines it writes are sized here, before any capture
        lbvh_note_write(ctx, d_triangles, (size_t)n * sizeof(lbvh_triangle));
        uint32_t* zero = nullptr;
        uint32_t zero_words = 0;
        lbvh_status src = (lbvh_status)lbvh_sort_scratch(ctx, capacity, &zero, &zero_words);
        if (src != LBVH_OK) return src;
        lbvh_fast_tri* lines = nullptr;
        if (fast_flag) {
            if ((src = ensure_fast_lines(ctx, n)) != LBVH_OK) return src;
            lines = ctx->fast_tris;
        }
        LBVH_REQUIRE(ctx, ((uintptr_t)d_triangles & 15) == 0 && ((uintptr_t)d_aabb & 15) == 0);
        lbvh_launch_animate_morton(ctx, *anim, const_cast<lbvh_triangle*>(d_triangles), n, capacity, h_box_min, h_box_max, d_keys, d_indices,
                                   d_aabb, zero, zero_words, lines, (flags & LBVH_BUILD_RESET_NODES) ? d_internal : nullptr,
                                   (flags & LBVH_BUILD_RESET_NODES) ? d_leaf : nullptr);
    }
    const bool morton_done = anim != nullptr;
    if (graphs && ctx->build_graph && ctx->build_graph_key == key) {
        LBVH_HIP_TRY(ctx, hipGraphLaunch(ctx->build_graph, ctx->stream));
        if (fast_flag) lbvh_note_fast_built(ctx, built);
        return LBVH_OK;
    }
    if (graphs && ctx->build_seen_key == key) {
        // second call with these arguments: every scratch buffer has its final size, so nothing allocates or
        // synchronises while the two lanes are recorded
        if (ctx->build_graph) { (void)hipGraphExecDestroy(ctx->build_graph); ctx->build_graph = nullptr; }
        int src = lbvh_ensure_side(ctx);
        if (src != LBVH_OK) return src;
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const lbvh_status rc = build_scene_enqueue(ctx, d_triangles, n, capacity, h_box_min, h_box_max, d_keys, d_indices, d_aabb,
                                                       d_internal, d_leaf, d_bvh, flags, morton_done);
            const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
            if (rc == LBVH_OK && e == hipSuccess && graph &&
                hipGraphInstantiate(&ctx->build_graph, graph, nullptr, nullptr, 0) == hipSuccess) {
                (void)hipGraphDestroy(graph);
                ctx->build_graph_key = key;
                LBVH_HIP_TRY(ctx, hipGraphLaunch(ctx->build_graph, ctx->stream));
                if (fast_flag) lbvh_note_fast_built(ctx, built);
                return LBVH_OK;
            }
            if (graph) (void)hipGraphDestroy(graph);
        }
        (void)hipGetLastError();
        ctx->build_graph = nullptr;
        ctx->build_graph_off = true;        // fall through to plain launches, now and from here on
        ctx->lane = 0;
        ctx->cur_stream = ctx->stream;
    }
    const lbvh_status rc = build_scene_enqueue(ctx, d_triangles, n, capacity, h_box_min, h_box_max, d_keys, d_indices, d_aabb,
                                               d_internal, d_leaf, d_bvh, flags, morton_done);
    // the key includes the scratch pointers as they are AFTER this call
    key = args_key;
    mix_scratch(mix);
    const uint64_t key2 = key ? key : 1;
    ctx->build_seen_key = rc == LBVH_OK ? key2 : 0;
    if (rc == LBVH_OK && fast_flag) lbvh_note_fast_built(ctx, built);
    return rc;
}

extern "C" {

lbvh_status lbvh_build_scene(lbvh_context* ctx, const lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                             const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                             lbvh_aabb* d_aabb, lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh,
                             uint32_t flags)
{
    return build_scene_impl(ctx, nullptr, d_triangles, n, capacity, h_box_min, h_box_max, d_keys, d_indices, d_aabb, d_internal, d_leaf,
                            d_bvh, flags);
}

lbvh_status lbvh_animate_build_scene(lbvh_context* ctx, const lbvh_triangle* d_rest, const uint32_t* d_body, const float* d_centres,
                                     float cos_angle, float sin_angle, lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                                     const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                                     lbvh_aabb* d_aabb, lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh,
                                     uint32_t flags)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, d_rest != nullptr && d_body != nullptr && d_centres != nullptr && d_triangles != nullptr);
    LBVH_REQUIRE(ctx, d_rest != d_triangles);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_rest & 15) == 0 && ((uintptr_t)d_triangles & 15) == 0 && ((uintptr_t)d_centres & 15) == 0);
    lbvh_anim anim = {d_rest, d_body, d_centres, cos_angle, sin_angle};
    return build_scene_impl(ctx, &anim, d_triangles, n, capacity, h_box_min, h_box_max, d_keys, d_indices, d_aabb, d_internal, d_leaf, d_bvh,
                            flags);
}

}  // extern "C"

static lbvh_status trace_impl(lbvh_context* ctx, const lbvh_camera* h_camera, int32_t x0, int32_t y0, int32_t x1,
                              int32_t y1, uint32_t shard_index, uint32_t shard_count, const lbvh_scene* h_scene,
                              int32_t mode, lbvh_hit* d_hits, lbvh_trace_stats* d_stats, uint32_t* d_tile_cost = nullptr,
                              bool packed = false)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, h_camera != nullptr && h_scene != nullptr);
    LBVH_REQUIRE(ctx, mode == LBVH_TRACE_REFERENCE || mode == LBVH_TRACE_FAST || mode == LBVH_TRACE_FAST_EXACT);
    const lbvh_camera cam = *h_camera;
    const lbvh_scene s = *h_scene;
    LBVH_REQUIRE(ctx, cam.screen_width > 0 && cam.screen_height > 0);
    LBVH_REQUIRE(ctx, x0 >= 0 && y0 >= 0 && x1 >= x0 && y1 >= y0);
    LBVH_REQUIRE(ctx, x1 <= cam.screen_width && y1 <= cam.screen_height);
    // new primary hits anywhere inside the path tracer's records (the frame, a rectangle of it, a share written at an offset): a new
    // frame for the live-path list of the last bounce (any overlap, not only the same base pointer)
    if (d_hits && ctx->ray_list.valid) {
        // (W x H records is an upper bound of every layout this call writes — frame, rectangle, packed share: dropping the list
        // once too often costs one scan of the states, nothing else)
        const uintptr_t w0 = (uintptr_t)d_hits, w1 = w0 + (size_t)cam.screen_width * (size_t)cam.screen_height * sizeof(lbvh_hit);
        const uintptr_t l0 = (uintptr_t)ctx->ray_list.hits, l1 = l0 + ctx->ray_list.count * sizeof(lbvh_hit);
        if (w0 < l1 && l0 < w1) ctx->ray_list.valid = false;
    }
    LBVH_REQUIRE(ctx, s.n >= 2);
    if (x1 == x0 || y1 == y0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_hits != nullptr && ((uintptr_t)d_hits & 15) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));

    trace_args a;
    a.cam = cam;
    a.x0 = x0; a.y0 = y0; a.x1 = x1; a.y1 = y1;
    a.shard_index = shard_index; a.shard_count = shard_count;
    a.line_bytes = 0;
    a.packed = packed ? 1u : 0u;
    a.ties = nullptr;
    a.tie_capacity = 0;
    a.sorted_indices = s.sorted_indices;
    a.tiles_x = (uint32_t)(x1 - x0 + 7) / 8;
    a.tiles_y = (uint32_t)(y1 - y0 + 7) / 8;
    const uint32_t n_tiles = shard_work(a.tiles_x * a.tiles_y, shard_index, shard_count);
    if (d_stats) LBVH_HIP_TRY(ctx, hipMemsetAsync(d_stats, 0, sizeof(lbvh_trace_stats), ctx->cur_stream));

    if (n_tiles == 0) return LBVH_OK;
    if (mode == LBVH_TRACE_REFERENCE) {
        LBVH_REQUIRE(ctx, s.sorted_indices && s.triangle_aabb && s.internal_nodes && s.leaf_nodes && s.bvh &&
                              s.triangles);
        if (d_stats)
            LBVH_LAUNCH(ctx, trace_reference_kernel<true>, dim3(n_tiles), dim3(64), a, s,
                               d_hits, d_stats);
        else
            LBVH_LAUNCH(ctx, trace_reference_kernel<false>, dim3(n_tiles), dim3(64), a, s,
                               d_hits, d_stats);
    } else {
        {
            const int frc = lbvh_require_fast(ctx, s, "lbvh_trace_primary (LBVH_TRACE_FAST)");
            if (frc != LBVH_OK) return frc;
        }
        const bool exact = mode == LBVH_TRACE_FAST_EXACT;
        if (exact) {
            // the list of tied candidates: one entry per ray of the launch (a scene of doubled triangles lists about one per hit
            // ray; when more coincide on most pixels the list runs over and the second resolve kernel takes those rays through the reference's loop)
            LBVH_REQUIRE(ctx, !d_stats && !d_tile_cost);
            LBVH_REQUIRE(ctx, s.sorted_indices && s.triangle_aabb && s.internal_nodes && s.leaf_nodes && s.bvh && s.triangles);
            LBVH_REQUIRE(ctx, x0 >= 0 && y0 >= 0 && x1 <= 65535 && y1 <= 65535);      // a listed ray's pixel is px | py << 16
            const uint64_t rays = (uint64_t)n_tiles * 64u;
            LBVH_REQUIRE(ctx, rays <= 0x3FFFFFFFull);
            const size_t had = ctx->tie_list_bytes;
            const int trc = lbvh_reserve(ctx, &ctx->tie_list, &ctx->tie_list_bytes, 16 + (size_t)rays * 16);
            if (trc != LBVH_OK) return trc;
            if (ctx->tie_list_bytes != had) ctx->tie_list_cleared = nullptr;      // a new block (possibly at the old address)
            if (ctx->tie_list != ctx->tie_list_cleared) {          // a fresh block (or a frame that did not reach its last kernel):
                LBVH_HIP_TRY(ctx, hipMemsetAsync(ctx->tie_list, 0, 16, ctx->cur_stream));      // both counters start at zero
                ctx->tie_turn = 0;
            }
            ctx->tie_list_cleared = nullptr;                       // valid again once this frame's last kernel is enqueued
            a.ties = (uint32_t*)ctx->tie_list;
            a.tie_capacity = (uint32_t)rays;
            a.tie_turn = ctx->tie_turn;
        }
        // 1 ray per lane = 8 x 8-pixel packets: 0.27 ms; 1x2: 0.43, 2x1: 0.45, 3x1: 0.68, 4x1: 0.82, 2x2: 0.92 ms
        // (more rays per lane cut node fetches per ray but lengthen every step and the per-tile critical path;
        // while the tile queues still cost 0.7 ms per launch, 2x1 had looked best).  4-wide 128-byte nodes
        // (the tree collapsed by one level, entries taken nearest first): 0.52x the steps but 0.37 ms, and
        // 110 us more build — re-measured after the queues were gone, still a loss.
        const int prc = launch_packets(ctx, a, d_hits, d_stats, d_tile_cost);
        if (prc != LBVH_OK) return prc;
        if (exact) {
            // two small launches: the lead among a record's candidates (compare-and-swap by the reference's visit order), then
            // the record as the reference computes it (or, after a list that ran over, the marked rays through the
            // reference's loop).  Without ties both find an empty list.
            LBVH_LAUNCH(ctx, resolve_ties_lead_kernel, dim3(64), dim3(256), a, s, d_hits);
            LBVH_LAUNCH(ctx, resolve_ties_write_kernel, dim3(256), dim3(64), a, s, n_tiles, d_hits);
            LBVH_HIP_TRY(ctx, hipGetLastError());
            ctx->tie_turn ^= 1u;
            ctx->tie_list_cleared = ctx->tie_list;
        }
    }
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

extern "C" {

lbvh_status lbvh_trace_primary(lbvh_context* ctx, const lbvh_camera* h_camera, int32_t x0, int32_t y0,
                               int32_t x1, int32_t y1, const lbvh_scene* h_scene, int32_t mode,
                               lbvh_hit* d_hits, lbvh_trace_stats* d_stats)
{
    return trace_impl(ctx, h_camera, x0, y0, x1, y1, 0, 1, h_scene, mode, d_hits, d_stats);
}

lbvh_status lbvh_trace_forget(lbvh_context* ctx)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    ctx->trace_history = false;
    ctx->trace_frame_valid = false;
    ctx->ray_list.valid = false;          // (and the path tracer's live-path list: the next bounce scans every state)
    return LBVH_OK;
}

lbvh_status lbvh_trace_costs_export(lbvh_context* ctx, uint32_t* d_frame_costs, uint32_t tiles_x, uint32_t tiles_y)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, d_frame_costs != nullptr && tiles_x > 0 && tiles_y > 0 && (uint64_t)tiles_x * tiles_y <= 0x7FFFFFFFull);
    // the layout of the last LBVH_TRACE_FAST trace: whole-frame traces and shards starting at the frame's origin
    // ... and only those: a sub-rectangle with the same tile counts but another origin would export its costs shifted
    if (!ctx->trace_history || ctx->trace_tiles_x != tiles_x || ctx->trace_tiles_y != tiles_y || ctx->trace_origin_x != 0 || ctx->trace_origin_y != 0)
        return lbvh_set_error(ctx, LBVH_ERR_INVALID_ARG, "lbvh_trace_costs_export",
                              "the last LBVH_TRACE_FAST trace was not a frame (or a shard of one) with this many tiles starting at pixel (0, 0)");
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n_work = ctx->trace_layout_work;
    const uint32_t* cost = (const uint32_t*)((const char*)ctx->trace_queues + 256);
    LBVH_LAUNCH(ctx, export_costs_kernel, dim3((n_work + 255) / 256), dim3(256), cost, n_work, tiles_x * tiles_y, ctx->trace_shard_index,
                ctx->trace_shard_count, d_frame_costs);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

lbvh_status lbvh_trace_costs_import(lbvh_context* ctx, const uint32_t* d_frame_costs, uint32_t tiles_x, uint32_t tiles_y)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, d_frame_costs != nullptr && tiles_x > 0 && tiles_y > 0 && (uint64_t)tiles_x * tiles_y <= 0x7FFFFFFFull);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = tiles_x * tiles_y;
    if (ctx->trace_frame_capacity < n) {
        LBVH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->trace_frame_costs) { LBVH_HIP_TRY(ctx, hipFree(ctx->trace_frame_costs)); ctx->trace_frame_costs = nullptr; }
        ctx->trace_frame_capacity = 0;
        ctx->trace_frame_valid = false;
        LBVH_HIP_TRY(ctx, hipMalloc((void**)&ctx->trace_frame_costs, (size_t)n * 4));
        ctx->trace_frame_capacity = n;
    }
    LBVH_HIP_TRY(ctx, hipMemcpyAsync(ctx->trace_frame_costs, d_frame_costs, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->cur_stream));
    ctx->trace_frame_tiles_x = tiles_x;
    ctx->trace_frame_tiles_y = tiles_y;
    ctx->trace_frame_valid = true;
    return LBVH_OK;
}

lbvh_status lbvh_trace_tile_costs(lbvh_context* ctx, const lbvh_camera* h_camera, const lbvh_scene* h_scene,
                                  lbvh_hit* d_hits, lbvh_trace_stats* d_stats, uint32_t* d_tile_steps)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, h_camera != nullptr && d_stats != nullptr && d_tile_steps != nullptr);
    return trace_impl(ctx, h_camera, 0, 0, h_camera->screen_width, h_camera->screen_height, 0, 1, h_scene, LBVH_TRACE_FAST,
                      d_hits, d_stats, d_tile_steps);
}

lbvh_status lbvh_trace_primary_shard(lbvh_context* ctx, const lbvh_camera* h_camera, uint32_t shard_index,
                                     uint32_t shard_count, const lbvh_scene* h_scene, int32_t mode,
                                     lbvh_hit* d_hits, lbvh_trace_stats* d_stats)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, h_camera != nullptr);
    LBVH_REQUIRE(ctx, shard_count >= 1 && shard_index < shard_count);
    return trace_impl(ctx, h_camera, 0, 0, h_camera->screen_width, h_camera->screen_height, shard_index, shard_count,
                      h_scene, mode, d_hits, d_stats);
}

lbvh_status lbvh_trace_primary_shard_packed(lbvh_context* ctx, const lbvh_camera* h_camera, uint32_t shard_index,
                                            uint32_t shard_count, const lbvh_scene* h_scene, int32_t mode,
                                            lbvh_hit* d_packed, lbvh_trace_stats* d_stats)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, h_camera != nullptr);
    LBVH_REQUIRE(ctx, shard_count >= 1 && shard_index < shard_count);
    return trace_impl(ctx, h_camera, 0, 0, h_camera->screen_width, h_camera->screen_height, shard_index, shard_count,
                      h_scene, mode, d_packed, d_stats, nullptr, true);
}

uint64_t lbvh_shard_records(int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count)
{
    if (width <= 0 || height <= 0 || shard_count == 0 || shard_index >= shard_count) return 0;
    const uint64_t tiles = (uint64_t)((width + 7) / 8) * (uint64_t)((height + 7) / 8);
    if (tiles > 0x7FFFFFFFull) return 0;
    return (uint64_t)shard_work((uint32_t)tiles, shard_index, shard_count) * 64u;
}

lbvh_status lbvh_frame_unpack(lbvh_context* ctx, const lbvh_hit* d_packed, uint64_t share_stride, uint32_t first_shard,
                              uint32_t n_shards, uint32_t shard_count, int32_t width, int32_t height, lbvh_hit* d_frame_hits)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, width > 0 && height > 0 && shard_count >= 1 && n_shards <= 65535u);
    LBVH_REQUIRE(ctx, (uint64_t)first_shard + n_shards <= shard_count);
    if (n_shards == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_packed != nullptr && d_frame_hits != nullptr && ((uintptr_t)d_packed & 15) == 0 && ((uintptr_t)d_frame_hits & 15) == 0);
    const uint64_t tiles64 = (uint64_t)((width + 7) / 8) * (uint64_t)((height + 7) / 8);
    LBVH_REQUIRE(ctx, tiles64 <= 0x7FFFFFFFull);
    const uint32_t n_tiles = (uint32_t)tiles64;
    uint32_t items = 0;          // the largest share of those handed in
    for (uint32_t s = 0; s < n_shards; s++) items = std::max(items, shard_work(n_tiles, first_shard + s, shard_count));
    LBVH_REQUIRE(ctx, share_stride >= (uint64_t)items * 64u);
    if (items == 0) return LBVH_OK;
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    LBVH_LAUNCH(ctx, frame_unpack_kernel, dim3((items + 3) / 4, n_shards), dim3(256), d_packed, share_stride, first_shard, shard_count, items,
                (uint32_t)((width + 7) / 8), n_tiles, (uint32_t)width, (uint32_t)height, d_frame_hits);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

}  // extern "C"
