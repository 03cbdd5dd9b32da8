// lbvh_common.h — shared definitions of the gfx950 LBVH library (context, error plumbing,
// wave64 primitives).  Everything here is CDNA4-only: 64-lane wavefronts are hard-coded.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/lbvh.h"
#include "../../include/lbvh_debug.h"

static_assert(sizeof(lbvh_aabb) == 32, "AABB must be 32 bytes (Sc/MeshBufferContainer.cs:103)");
static_assert(sizeof(lbvh_triangle) == 128, "Triangle must be 128 bytes (Sc/MeshBufferContainer.cs:98)");
static_assert(sizeof(lbvh_internal_node) == 24, "InternalNode must be 24 bytes");
static_assert(sizeof(lbvh_leaf_node) == 8, "LeafNode must be 8 bytes");
static_assert(sizeof(lbvh_hit) == 16, "hit record must be 16 bytes");

#define LBVH_WAVE 64

// device-side fault codes (ctx->fault_host): a bounded spin of an inter-workgroup protocol gave up
#define LBVH_FAULT_SORT_LOOKBACK 1u
// a per-ray traversal stack ran out of entries (lbvh_trace_rays / lbvh_path_bounce): hits of that launch are invalid
#define LBVH_FAULT_RAY_STACK 2u
// lbvh_frame_wait gave up: another rank's completion flag for this frame never arrived
#define LBVH_FAULT_FRAME_WAIT 3u

// polls before a spin gives up: each poll is a round trip to the coherence point (>= 0.5 us), so this is seconds —
// orders of magnitude beyond any legitimate wait (a predecessor tile's run time), and never a hung GPU
#define LBVH_SPIN_LIMIT (1u << 22)

// Derived traversal node for LBVH_TRACE_FAST: both child boxes + child references, 64 bytes,
// one 64-B aligned fetch per traversal step.  Child reference: an index into the array of 64-byte lines that
// holds the nodes and, from index fast_leaf_base on, the sorted triangles; bit 31 set = leaf.
struct alignas(64) lbvh_fast_node {
    float lmin[3]; uint32_t left;
    float lmax[3]; uint32_t right;
    float rmin[3]; uint32_t pad0;
    float rmax[3]; uint32_t pad1;
};
static_assert(sizeof(lbvh_fast_node) == 64, "fast node must be 64 bytes");

// Triangle line for LBVH_TRACE_FAST, one per triangle in the caller's ORIGINAL order (measured: sorted or original
// order of these lines makes no difference to the traversal, 0.226 ms either way, and original order lets the Morton
// kernel write them from the positions it has in registers — no gather of the 128-byte records after the sort):
// first vertex and the two edge vectors e1 = b - a, e2 = c - a (the fp32 differences the intersection test starts
// with, Raytracing.compute:41-42, taken once at build time) and the triangle's index (the hit record's
// triangleIndex).  A 64-byte line in the SAME allocation as the traversal nodes, right behind them: a child reference
// (node index, or LEAF | leaf_base + triangle index) is one index into one array of 64-byte lines, and a line fetch
// is base + (index << 6) whatever it points at.
// Layout: the packet walk fetches node lines with dwords k and k + 4 (k = 0, 1, 2 and 8, 9, 10) swapped for the axes
// its rays travel down (lbvh_trace.hip), and fetches a child's line before it knows — per lane — what the child
// is; so the triangle line reads the same under any such swap: a and e1 are stored twice, e2 and the index sit in
// the four dwords 3, 7, 11, 15 the swap never touches.
struct alignas(64) lbvh_fast_tri {
    float a[3]; uint32_t orig_index;
    float a_again[3]; float e2x;
    float e1[3]; float e2y;
    float e1_again[3]; float e2z;
};
static_assert(sizeof(lbvh_fast_tri) == 64, "fast triangle must be 64 bytes");

struct lbvh_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // radix sort scratch (ping-pong pairs + per-tile digit tables), grown lazily
    void* sort_scratch = nullptr;
    size_t sort_scratch_bytes = 0;
    // Two lanes: lane 0 = the context's stream; lane 1 = an internal side stream on which lbvh_build_scene
    // builds the derived traversal scene while lane 0 builds the reference's arrays (independent chains
    // of latency-bound kernels).  Launches go to cur_stream; scan / refit scratch exists once per lane.
    hipStream_t side_stream = nullptr;
    hipStream_t cur_stream = nullptr;
    int lane = 0;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_hier = nullptr;          // the range hierarchy of the build in flight is complete
    // range hierarchy of the sorted leaf boxes (lbvh_build.hip): level 0 = the leaf AABBs in sorted order
    void* hier = nullptr;
    size_t hier_bytes = 0;
    // lbvh_build_scene replays a captured hipGraph when it is called again with the same arguments (per-frame
    // rebuilds): ~20 short dependent kernels on two streams, launch gaps included, become one graph launch
    hipGraphExec_t build_graph = nullptr;
    uint64_t build_graph_key = 0;     // arguments the graph was captured for
    uint64_t build_seen_key = 0;      // arguments of the last plain call (scratch is sized for them)
    bool build_graph_off = false;     // capture failed once: stay on plain launches
    // distribute-keys / aligned-keys scan scratch
    void* scan_scratch[2] = {nullptr, nullptr};
    size_t scan_scratch_bytes[2] = {0, 0};
    // refit scratch: frontier list + range levels (the reference's atomicsData has no counterpart)
    uint32_t* refit_scratch[2] = {nullptr, nullptr};
    size_t refit_scratch_words[2] = {0, 0};
    // derived fast-traversal scene
    lbvh_fast_node* fast_nodes = nullptr;
    lbvh_fast_tri* fast_tris = nullptr;    // = fast_nodes + fast_capacity (same allocation)
    uint32_t fast_capacity = 0;            // = the line index of sorted triangle 0 (leaf base)
    // what the derived scene was built FROM: LBVH_TRACE_FAST / lbvh_trace_rays accept a scene only if it names these
    // buffers and no library call has written into them since (lbvh_note_write) — the reference's Dispatch binds its
    // buffers statelessly (Sc/RaytracingMeshDrawer.cs:65-70), so a stale cache must never answer for a scene
    struct { const void *triangles, *sorted_indices, *triangle_aabb; uint32_t n; } fast_src = {nullptr, nullptr, nullptr, 0};
    bool fast_valid = false;
    // sort: 8 per-XCD ticket queues only on the layout they were designed for (all 256 CUs of an SPX device behind an
    // unmasked stream: workgroups dealt round-robin over the XCDs); anything else takes tiles in ticket order
    uint32_t sort_queues = 1;
    uint32_t sort_hint_streak = 0;          // consecutive lbvh_launch_sort calls that found the two-level form's hint in order
    uint32_t sort_queues_detected = 1;      // what lbvh_create found (lbvh_debug_switch(LBVH_DEBUG_SORT_QUEUES, 0) goes back to it)
    // lbvh_debug_switch (include/lbvh_debug.h): all 0 in the product
    uint32_t debug_switch[LBVH_DEBUG_SWITCHES] = {};
    // device-side protocol faults (a bounded spin gave up): one mapped host word, checked by lbvh_sync / download.  The same
    // 256-byte mapped block carries, in words 16 .. 19, the largest (balanced) bucket of the last sort of 2^15 .. 2^21 pairs tagged with its fine shift, and in word 24 a work estimate of the last but one traced frame (written by the
    // first pass kernel, read without synchronisation by the next lbvh_launch_sort: which form to take — lbvh_sort.hip)
    uint32_t* fault_host = nullptr;
    uint32_t* fault_dev = nullptr;
    // packet traversal scheduling: step count of every tile in the last trace + the dispatch order made from it
    void* trace_queues = nullptr;
    size_t trace_queues_bytes = 0;
    uint64_t trace_layout = 0;      // frame layout (tiles, shard, origin) the history belongs to
    uint32_t trace_layout_work = 0;
    uint32_t trace_shard_index = 0, trace_shard_count = 1, trace_tiles_x = 0, trace_tiles_y = 0;   // of that trace
    int32_t trace_origin_x = 0, trace_origin_y = 0;                                                // ... and its rectangle's origin
    bool trace_history = false;
    float fast_centre[3] = {0.0f, 0.0f, 0.0f};   // centre of the scene box the derived scene was built with
    uint32_t trace_counts_turn = 0;  // which of the two class-counter sets the next filing counts into
    lbvh_camera trace_camera = {};   // camera of the trace the history was recorded under
    // multi-GPU frames: every rank's tile costs of the last frame, merged by the caller (lbvh_trace_costs_import): where a
    // tile of a moved camera came from usually belongs to another rank
    uint32_t* trace_frame_costs = nullptr;
    uint32_t trace_frame_tiles_x = 0, trace_frame_tiles_y = 0;
    uint32_t trace_frame_capacity = 0;
    bool trace_frame_valid = false;
    // the traversal tree of the derived scene (aligned keys, own topology and boxes)
    void* fast_tree = nullptr;
    size_t fast_tree_bytes = 0;

    // lbvh_trace_rays: live-ray list
    void* ray_scratch = nullptr;
    // the live-path list the last lbvh_path_bounce left behind (lbvh_path.hip): the next bounce of the same frame starts from it
    // instead of scanning every pixel; dropped by anything that writes the states / hit records in between (lbvh_note_write)
    struct { const void *states, *hits; size_t count; uint32_t bounce, turn; bool valid; } ray_list = {nullptr, nullptr, 0, 0, 0, false};
    uint32_t ray_stack_lds = 16;              // lbvh_debug_ray_stack_split
    uint32_t ray_stack_deep = 0xFFFFFFFFu;    // lbvh_debug_ray_stack_limit: entries of the device-memory part the walkers may use
    size_t ray_scratch_bytes = 0;
    // the derived scene as four-wide nodes for the per-ray walk: made by the first lbvh_trace_rays / bounce after a rebuild
    void* wide_nodes = nullptr;
    size_t wide_nodes_bytes = 0;
    bool wide_valid = false;
    // LBVH_TRACE_FAST_EXACT: rays listed for the reference's walk (lbvh_trace.hip retrace_ties_kernel)
    void* tie_list = nullptr;
    size_t tie_list_bytes = 0;
    void* tie_list_cleared = nullptr;         // the block whose two counters are in rotation
    uint32_t tie_turn = 0;
    lbvh_ray_stats* ray_stats = nullptr;      // lbvh_ray_stats_target: the four-wide walkers add their counters here while set
    uint32_t ray_walker = 1;                  // lbvh_debug_ray_walker: 0 binary nodes, 1 four-wide, 2 four-wide with the few-rays kernel always

    // per-kernel event profiling (lbvh_profile_begin / lbvh_profile_end)
    struct prof_span { const char* name; hipEvent_t a, b; };
    bool prof_enabled = false;
    std::vector<prof_span> prof_spans;
    std::vector<hipEvent_t> prof_pool;
};

hipEvent_t lbvh_prof_event(lbvh_context* ctx);

// Every kernel launch of the library goes through this: a plain launch on the context's current lane,
// bracketed by two events when profiling is on.
#define LBVH_LAUNCH(ctx, kernel, grid, block, ...)                                      \
    do {                                                                                \
        hipEvent_t _a = nullptr, _b = nullptr;                                          \
        if ((ctx)->prof_enabled) {                                                      \
            _a = lbvh_prof_event(ctx);                                                  \
            _b = lbvh_prof_event(ctx);                                                  \
            (void)hipEventRecord(_a, (ctx)->cur_stream);                                \
        }                                                                               \
        hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->cur_stream, __VA_ARGS__);     \
        if ((ctx)->prof_enabled) {                                                      \
            (void)hipEventRecord(_b, (ctx)->cur_stream);                                \
            (ctx)->prof_spans.push_back({#kernel, _a, _b});                             \
        }                                                                               \
    } while (0)

int lbvh_set_error(lbvh_context* ctx, int code, const char* what, const char* detail);

#define LBVH_HIP_TRY(ctx, expr)                                                      \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess)                                                        \
            return lbvh_set_error((ctx), _e == hipErrorOutOfMemory ? LBVH_ERR_OUT_OF_MEMORY \
                                                                   : LBVH_ERR_HIP,    \
                                  #expr, hipGetErrorString(_e));                     \
    } while (0)

#define LBVH_REQUIRE(ctx, cond)                                                      \
    do {                                                                             \
        if (!(cond)) return lbvh_set_error((ctx), LBVH_ERR_INVALID_ARG, "invalid argument", #cond); \
    } while (0)

// Stage launchers shared between translation units (validated arguments, no error reporting of
// their own beyond hipGetLastError at the caller).
// d_zero_word (may be nullptr): a word the tree kernel clears on the way — the counter of the refit that follows
int lbvh_launch_tree(lbvh_context* ctx, uint32_t n, const uint32_t* d_keys, lbvh_internal_node* d_internal,
                     lbvh_leaf_node* d_leaf, uint32_t* d_zero_word);
// Morton / AABB kernel that also clears d_zero[0 .. zero_words) (the scratch of the sort that follows), with
// d_lines writes the derived scene's 64-byte triangle lines (original order), and with d_reset_internal / d_reset_leaf refills
// what LBVH_BUILD_RESET_NODES has to refill of the node arrays: the slots past the tree and the root's parent word
int lbvh_launch_morton(lbvh_context* ctx, const lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                       const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                       lbvh_aabb* d_aabb, uint32_t* d_zero, uint32_t zero_words, lbvh_fast_tri* d_lines,
                       lbvh_internal_node* d_reset_internal = nullptr, lbvh_leaf_node* d_reset_leaf = nullptr);
// lbvh_animate fused into the Morton kernel (lbvh_animate_build_scene): the rest pose is moved into d_triangles on the way
struct lbvh_anim { const lbvh_triangle* rest; const uint32_t* body; const float* centres; float cos_angle, sin_angle; };
int lbvh_launch_animate_morton(lbvh_context* ctx, const lbvh_anim& anim, lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                               const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                               lbvh_aabb* d_aabb, uint32_t* d_zero, uint32_t zero_words, lbvh_fast_tri* d_lines,
                               lbvh_internal_node* d_reset_internal = nullptr, lbvh_leaf_node* d_reset_leaf = nullptr);
// lbvh_build_scene (LBVH_BUILD_FAST_SCENE) after the sort as three merged launches on the current stream; *done = false: too large,
// nothing enqueued
bool lbvh_post_sort_merges(uint32_t n);
int lbvh_launch_post_sort_merged(lbvh_context* ctx, uint32_t n, uint32_t* d_keys, const lbvh_aabb* d_triangle_aabb,
                                 const uint32_t* d_sorted_indices, const float box_min[3], const float box_max[3],
                                 uint32_t* d_aligned_keys, lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh,
                                 lbvh_fast_node* d_fused, uint32_t leaf_base, bool* done);
// the sort with its scratch described / already cleared by the caller
int lbvh_sort_scratch(lbvh_context* ctx, uint32_t count, uint32_t** d_zero, uint32_t* zero_words);
// key_bits: keys below 2^key_bits are the common case (30 for Morton codes; anything above still sorts correctly): where the
// two-level form (lbvh_sort.hip) takes its bucket digit from
int lbvh_launch_sort(lbvh_context* ctx, uint32_t* d_keys, uint32_t* d_values, uint32_t count, bool scratch_cleared, uint32_t key_bits = 32u);
// the frontier counter lbvh_launch_refit(n) will use on the current lane (sizes the scratch)
int lbvh_refit_counter(lbvh_context* ctx, uint32_t n, uint32_t** d_counter);
// the stand-alone refit (lbvh_refit): d_sorted_indices may be nullptr (boxes already in leaf order)
int lbvh_launch_refit(lbvh_context* ctx, uint32_t n, const lbvh_internal_node* d_internal, const lbvh_leaf_node* d_leaf,
                      const lbvh_aabb* d_triangle_aabb, const uint32_t* d_sorted_indices, lbvh_aabb* d_bvh, bool counter_cleared);
// The refit of lbvh_build_scene / lbvh_build_fast_scene is a range query inside the tree kernel (lbvh_build.hip, "range
// hierarchy"): lbvh_launch_gather_hier writes the leaf boxes in sorted order + the union of every aligned group of
// 2^k leaves into the context's hierarchy (and, with d_aligned_keys_out, the derived tree's aligned keys
// k'_i = i + max_{j<=i}(morton(centre of leaf box j) - j)); the two tree launchers below read it.
int lbvh_hier_reserve(lbvh_context* ctx, uint32_t n);      // sizes the hierarchy (no launch): before a graph capture
// with_top = false leaves the levels above 1024 leaves to a later lbvh_launch_hier_top
int lbvh_launch_gather_hier(lbvh_context* ctx, uint32_t n, const lbvh_aabb* d_triangle_aabb, const uint32_t* d_sorted_indices,
                            const float box_min[3], const float box_max[3], uint32_t* d_aligned_keys_out, bool with_top);
int lbvh_launch_hier_top(lbvh_context* ctx, uint32_t n);
// TreeConstructor + BVHData in one kernel (the reference's arrays)
int lbvh_launch_tree_boxes(lbvh_context* ctx, uint32_t n, const uint32_t* d_keys, lbvh_internal_node* d_internal,
                           lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh);
// the derived traversal tree: 64-byte traversal nodes straight from the keys (leaf references
// LEAF | leaf_base + ORIGINAL triangle index, looked up in d_sorted_indices)
int lbvh_launch_tree_fused(lbvh_context* ctx, uint32_t n, const uint32_t* d_keys, const uint32_t* d_sorted_indices,
                           lbvh_fast_node* d_fused, uint32_t leaf_base);

// A library call is about to write [p, p + bytes): if that touches what the derived scene was built from, the
// derived scene is stale from here on.
void lbvh_note_write(lbvh_context* ctx, const void* p, size_t bytes);
// the derived scene now describes `s` (called where lbvh_build_fast_scene's work is enqueued or replayed)
void lbvh_note_fast_built(lbvh_context* ctx, const lbvh_scene& s);
// LBVH_OK iff the derived scene was built from exactly these buffers and they have not been written since
int lbvh_require_fast(lbvh_context* ctx, const lbvh_scene& s, const char* who);
// non-zero fault word -> LBVH_ERR_HIP with the code in the message (after a stream sync)
int lbvh_check_fault(lbvh_context* ctx);

// Creates the side stream and its fork / join events on first use.
int lbvh_ensure_side(lbvh_context* ctx);

// Grow-only scratch helper: (re)allocates *ptr to at least `bytes`.
int lbvh_reserve(lbvh_context* ctx, void** ptr, size_t* have, size_t bytes);

// ---- wave64 device primitives --------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// which of the 8 XCDs (each with its own L2) this wave runs on: s_getreg_b32 HW_REG_XCC_ID (id 20), bits [3:0]
__device__ __forceinline__ uint32_t xcc_id()
{
    return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u;
}

// number of set bits of `mask` in lanes below the calling lane (v_mbcnt_lo + v_mbcnt_hi)
__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// DPP controls (gfx9 family): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_step(uint32_t v)
{
    // old = 0 (identity) for lanes whose source is out of range / masked off
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

// Inclusive prefix sum across the 64 lanes of a wave: 4 row_shr steps inside each 16-lane row,
// then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (the gfx9 wave64 scan).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    v = dpp_add_step<0x111, 0xf>(v);
    v = dpp_add_step<0x112, 0xf>(v);
    v = dpp_add_step<0x114, 0xf>(v);
    v = dpp_add_step<0x118, 0xf>(v);
    v = dpp_add_step<0x142, 0xa>(v);
    v = dpp_add_step<0x143, 0xc>(v);
    return v;
}

__device__ __forceinline__ uint32_t wave_exclusive_sum(uint32_t v) { return wave_inclusive_sum(v) - v; }

// wave total, valid in every lane (lane 63 of the inclusive scan, read back as a scalar)
__device__ __forceinline__ uint32_t wave_total_from_inclusive(uint32_t inclusive)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)inclusive, 63);
}

#if defined(__HIPCC__)
// streaming (non-temporal) 16-byte accesses: data read or written once by a kernel, not to be kept in the caches
typedef float lbvh_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lbvh_nt_load(const float4* p)
{
    const lbvh_f4v v = __builtin_nontemporal_load(reinterpret_cast<const lbvh_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lbvh_nt_store(float4* p, const float4 v)
{
    const lbvh_f4v w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<lbvh_f4v*>(p));
}
#endif
