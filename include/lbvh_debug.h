/* lbvh_debug.h — test hooks and measurement aids of liblbvh.so.
 *
 * NOT part of the drop-in boundary: nothing here replaces a reference call site, and a Unity host never needs
 * it.  The entry points live in the same library so that the tests, bench.py and tools/ can reach into a context; they follow
 * the same conventions as include/lbvh.h (plain C types, lbvh_status, no exceptions).  bindings/csharp/LbvhNativeDebug.cs is the
 * matching P/Invoke table. */
#ifndef LBVH_DEBUG_H
#define LBVH_DEBUG_H

#include "lbvh.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Switches that used to be environment variables read inside the library (rounds 1 - 4); all default to 0 = the product's
 * behaviour, all are per context, none changes a result.
 *   LBVH_DEBUG_SORT_QUEUES      1 / 8: force the sort's single ticket queue / the eight per-XCD queues (0: what lbvh_create
 *                               detected — eight only on the whole 256-CU device behind an unmasked stream)
 *   LBVH_DEBUG_COLD_ORDER       1: frames without history take their tiles row-major (round 3) instead of centre-out
 *   LBVH_DEBUG_BUILD_FORM       how lbvh_build_scene enqueues its chain: 0 as shipped (merged plain launches up to 2 M
 *                               triangles, a replayed two-stream graph above), 1 plain launches always, 2 a replayed graph
 *                               always, 3 the two-stream form (plain), 4 the two-stream form as a graph
 *   LBVH_DEBUG_FRAME_WAIT_MS    wall-clock bound of lbvh_frame_wait's device-side wait in milliseconds (0: the default, 20 s)
 *   LBVH_DEBUG_SORT_FORM        1: the sort always runs the four 8-bit LSD passes, 2: the two-level form wherever the size allows
 *                               (0: chosen per call from the last sorts' largest buckets)
 *   LBVH_DEBUG_FAIL_RESERVE     k > 0: the k-th growth of a context-owned scratch buffer from now on fails as hipMalloc does when
 *                               the device is full (LBVH_ERR_OUT_OF_MEMORY; the old block is gone, the pointer is null), then the
 *                               switch is 0 again — the error paths of the sort / build / trace entry points on a healthy box */
enum {
    LBVH_DEBUG_SORT_QUEUES = 0,
    LBVH_DEBUG_COLD_ORDER = 1,
    LBVH_DEBUG_BUILD_FORM = 2,
    LBVH_DEBUG_FRAME_WAIT_MS = 3,
    LBVH_DEBUG_SORT_FORM = 4,
    LBVH_DEBUG_FAIL_RESERVE = 5,
    LBVH_DEBUG_SWITCHES = 6
};
lbvh_status lbvh_debug_switch(lbvh_context* ctx, uint32_t which, uint32_t value);

/* Host-side view of the sort's tile hand-out order (pure function, no GPU): the tile that the k-th ticket of queue x
 * stands for.  A pass kernel takes its tile from atomic tickets; on the full 8-XCD device there are 8 queues (one per
 * XCD, `group` consecutive tiles each in turn: neighbouring tiles meet in one L2), on anything else a single queue
 * (tile = ticket).  Exposed so the order invariant the decoupled look-back relies on — every tile below a handed-out
 * tile has been handed out or is the next ticket of some queue — can be model-checked without a GPU. */
uint32_t lbvh_debug_sort_ticket_tile(uint32_t k, uint32_t x, uint32_t group, uint32_t queues);

/* Test hook: how many of a ray's stack entries live in LDS (1..16, default 16) before the walker of lbvh_trace_rays /
 * lbvh_path_bounce spills to its device-memory slab.  Results do not depend on it; tests lower it so the deep part
 * of the stack is exercised by ordinary scenes. */
lbvh_status lbvh_debug_ray_stack_split(lbvh_context* ctx, uint32_t lds_entries);

/* Measurement aid (cfg5's roofline): while d_stats is non-NULL, every launch of the four-wide per-ray walk (lbvh_trace_rays,
 * lbvh_path_bounce, lbvh_path_first_bounce) ADDS what it did to it: rays walked, 128-byte four-wide node lines fetched (one per
 * ray-step), triangle lines fetched and tested.  Zero it yourself; NULL switches the counting off (the default: the counting
 * kernels are separate instantiations, the product's carry none of it). */
typedef struct lbvh_ray_stats {
    uint64_t rays;
    uint64_t node_fetches;
    uint64_t triangle_tests;
} lbvh_ray_stats;
lbvh_status lbvh_ray_stats_target(lbvh_context* ctx, lbvh_ray_stats* d_stats);

/* Test hook: how many entries of the device-memory part of a ray's stack the walkers may use (0 = all of it: 48 for the
 * binary walk, 112 for the four-wide one — more than any tree of this library can ask for).  A stack that runs out does not
 * drop the entry silently: the launch sets the context's fault word and the next lbvh_sync / lbvh_buffer_download returns
 * LBVH_ERR_HIP ("a per-ray traversal stack ran out of entries").  Tests lower the limit to see exactly that. */
lbvh_status lbvh_debug_ray_stack_limit(lbvh_context* ctx, uint32_t deep_entries);

/* Test hook: which walk lbvh_trace_rays / lbvh_path_bounce run — 1 (default): four-wide nodes (each binary node with its
 * largest children opened, made on first use after a rebuild; from bounce 1 on lbvh_path_bounce takes the kernel that keeps
 * a step's two fetches in flight at once: few live rays, the launch is the chain of its longest), 2: that kernel for every
 * launch, 0: the binary nodes the packet walk uses.  Hit records do not depend on it (ties go to the lower triangle index
 * on all three). */
lbvh_status lbvh_debug_ray_walker(lbvh_context* ctx, uint32_t walker);

/* Profiling aid: one LBVH_TRACE_FAST frame that also records, per 8x8-pixel tile (row-major,
 * ceil(W/8) x ceil(H/8) entries), the number of node fetches its packet needed. */
lbvh_status lbvh_trace_tile_costs(lbvh_context* ctx, const lbvh_camera* h_camera, const lbvh_scene* h_scene,
                                  lbvh_hit* d_hits, lbvh_trace_stats* d_stats, uint32_t* d_tile_steps);

/* Per-kernel timing: between lbvh_profile_begin and lbvh_profile_end every kernel the library
 * launches on this context is bracketed by its own pair of HIP events (on the context's stream).
 * lbvh_profile_end waits for the stream and returns one row per kernel name: launches and the
 * summed device time.  Off by default: it adds two event records per launch, so it is never on
 * inside a throughput measurement. */
typedef struct lbvh_profile_row {
    char     name[48];
    uint32_t launches;
    float    total_ms;
} lbvh_profile_row;
lbvh_status lbvh_profile_begin(lbvh_context* ctx);
lbvh_status lbvh_profile_end(lbvh_context* ctx, lbvh_profile_row* h_rows, int32_t max_rows,
                             int32_t* out_rows);

/* The shader clock the chip holds under a vector-ALU-bound load (MHz): every CU runs dependent fp32 work for a few
 * hundred microseconds while each wave reads the shader-cycle counter (s_memtime) and the constant 100 MHz counter
 * (s_memrealtime) before and after; the median ratio is returned.  Blocking.  Issue-rate figures (instructions per
 * cycle x this clock) use it instead of assuming the 2.4 GHz maximum. */
lbvh_status lbvh_clock_probe(lbvh_context* ctx, float* out_shader_mhz);

/* Streaming device-to-device copy of `bytes` (float4 per lane) on the context's stream: the
 * box's own HBM copy rate is the measured roofline denominator quoted beside the 8 TB/s spec. */
lbvh_status lbvh_copy_bandwidth_probe(lbvh_context* ctx, void* d_dst, const void* d_src,
                                      size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* LBVH_DEBUG_H */
