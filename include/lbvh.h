/*
 * lbvh.h — C ABI of liblbvh.so, the MI355X (gfx950) native LBVH ray-tracing hot path.
 *
 * This is the drop-in boundary for the hot path of drzhn/UnitySimpleRaytracing:
 *   Morton/AABB -> radix sort -> DistributeKeys -> Karras tree -> AABB refit -> primary-ray traversal.
 * In the reference that path sits behind UnityEngine.ComputeBuffer / ComputeShader.Dispatch; here
 * every Dispatch (and the two CPU loops the reference runs on the host) is one `extern "C"` call
 * that a C# [DllImport] shim, the C++ host classes under unitysimpleraytracing_amd/host/ and the
 * Python ctypes tests all bind the same way.  Citations are file:line in the reference tree
 * (Sc/ = Assets/_Scripts/, Sh/ = Assets/_Shaders/).
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary; every call returns lbvh_status
 *     (0 = LBVH_OK, negative = error; lbvh_last_error(ctx) gives the text).
 *   - pointers named d_* are DEVICE pointers (hipMalloc'ed memory on the context's GPU: from
 *     lbvh_buffer_alloc, or any other allocator, e.g. a torch tensor's data_ptr()).
 *     Pointers named h_* are host pointers borrowed for the duration of the call.
 *   - all stage calls are asynchronous and ordered on the context's HIP stream;
 *     lbvh_buffer_download and lbvh_sync block (= ComputeBuffer.GetData, Sc/DataBuffer.cs:50-54).
 *   - a context is bound to one GPU and is not thread-safe; multi-GPU = one context per device
 *     (one process per GPU in bench.py; one process driving N contexts in host/lbvh_host.hpp MultiGpuDrawer).
 *   - buffer layouts are the reference's, bit for bit (Sh/Constants.cginc:9-54,
 *     Sc/SceneDataTypes.cs:4-89).
 */
#ifndef LBVH_H
#define LBVH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBVH_ABI_VERSION 11

/* ---- status codes ------------------------------------------------------------------------- */
typedef int32_t lbvh_status;
#define LBVH_OK                 0
#define LBVH_ERR_INVALID_ARG   -1   /* null pointer, n < 2, n > capacity, bad tile rectangle ...   */
#define LBVH_ERR_OUT_OF_MEMORY -2
#define LBVH_ERR_HIP           -3   /* a HIP runtime call failed, or a device-side wait gave up (the next lbvh_sync /
                                       lbvh_buffer_download reports it ONCE: what was enqueued before that call is
                                       invalid, later work is judged on its own); see lbvh_last_error  */
#define LBVH_ERR_NO_DEVICE     -4   /* no usable gfx950 device                                     */

/* ---- scene structs: the reference's buffer element layouts --------------------------------- */

/* Sh/Constants.cginc:9-15, Sc/SceneDataTypes.cs:4-16 — 32 bytes */
typedef struct lbvh_aabb {
    float min[3];
    float _dummy0;
    float max[3];
    float _dummy1;
} lbvh_aabb;

/* Sh/Constants.cginc:36-54, Sc/SceneDataTypes.cs:18-41 — 128 bytes */
typedef struct lbvh_triangle {
    float a[3];        float _dummy0;
    float b[3];        float _dummy1;
    float c[3];        float _dummy2;
    float a_uv[2];
    float b_uv[2];
    float c_uv[2];
    float _dummy3[2];
    float a_normal[3]; float _dummy4;
    float b_normal[3]; float _dummy5;
    float c_normal[3]; float _dummy6;
} lbvh_triangle;

/* Sh/Constants.cginc:17-18 */
#define LBVH_INTERNAL_NODE 0u
#define LBVH_LEAF_NODE     1u
/* every word of an unwritten node slot (Sc/SceneDataTypes.cs:63-71,85-89); also root.parent */
#define LBVH_NULL          0xFFFFFFFFu

/* Sh/Constants.cginc:20-28, Sc/SceneDataTypes.cs:43-72 — 24 bytes */
typedef struct lbvh_internal_node {
    uint32_t leftNode;
    uint32_t leftNodeType;
    uint32_t rightNode;
    uint32_t rightNodeType;
    uint32_t parent;
    uint32_t index;
} lbvh_internal_node;

/* Sh/Constants.cginc:30-34, Sc/SceneDataTypes.cs:74-89 — 8 bytes */
typedef struct lbvh_leaf_node {
    uint32_t parent;
    uint32_t index;
} lbvh_leaf_node;

/* MAX_FLOAT is the INTEGER literal 0x7F7FFFFF converted to float (Sh/Constants.cginc:7):
 * 2139095040.0f, not FLT_MAX.  A miss carries exactly this value in lbvh_hit.t. */
#define LBVH_MAX_FLOAT 2139095040.0f

/* Per-ray result = the reference's RaycastResult (Sh/Raytracing/Raytracing.compute:30-35)
 * in its field order: distance, triangleIndex, uv.  16 bytes.
 * (The reference consumes it in-kernel for shading, :178-184; here it is the kernel output.) */
typedef struct lbvh_hit {
    float    t;      /* distance; LBVH_MAX_FLOAT on a miss                       */
    uint32_t tri;    /* ORIGINAL triangle index (into triangleData); 0 on a miss */
    float    u, v;   /* barycentrics of b and c; (0,0) on a miss                 */
} lbvh_hit;

/* The uniforms RaytracingMeshDrawer.Update sets each frame (Sc/RaytracingMeshDrawer.cs:78-81)
 * plus the implicit _ProjectionParams.y (camera near plane) the kernel reads
 * (Sh/Raytracing/Raytracing.compute:108). */
typedef struct lbvh_camera {
    int32_t screen_width;        /* screenWidth                                               */
    int32_t screen_height;       /* screenHeight                                              */
    float   camera_fov;          /* cameraFov = tan(fovY/2), already the tangent              */
    float   near_plane;          /* _ProjectionParams.y                                       */
    float   camera_to_world[16]; /* cameraToWorldMatrix, row-major m00,m01,m02,m03,m10,...    */
} lbvh_camera;

/* Traversal flavours of lbvh_trace_primary. */
#define LBVH_TRACE_REFERENCE 0  /* the reference's visit order, no pruning, separate node arrays */
#define LBVH_TRACE_FAST      1  /* 8x8 packets over fused 64-B nodes, near-first, t-pruned; same min-t (see the note below) */
#define LBVH_TRACE_FAST_EXACT 2 /* LBVH_TRACE_FAST with the reference's choice on exact ties: every record equals, word for
                                 * word, the record of the reference's loop under the accept rule of the note below (the
                                 * oracle's orc_trace_primary_rule with fast_rule = 1) — that is LBVH_TRACE_REFERENCE's record
                                 * at every pixel whose reference winner is not a t in front of its own triangle's box
                                 * (every pixel of every scene in the test suite but one, which is pinned:
                                 * tests/golden/grazing_ray_case.json).  A ray that meets two triangles at EXACTLY the same t
                                 * (the case in which the fast walk's order-independent choice — lowest triangle index — can
                                 * differ from the triangle the reference's visit order meets first) is given to the triangle
                                 * that order meets first: the order is read off the scene's internalNodes / leafNodes
                                 * (parent words, child types; leaf j's `index` is j, as TreeConstructor writes it,
                                 * BVH.compute:116-120).  No d_stats with this mode. */
/* Where both fast modes differ from LBVH_TRACE_REFERENCE — by a rule, not by chance (tests/test_grazing_ray.py).
 * The reference prunes nothing, so its record is the minimum COMPUTED t over every triangle whose box the ray's line passes —
 * including, for a ray within ~0.01 degree of a triangle's plane, a t that is noise of the fp32 triangle test
 * (Raytracing.compute:37-73: det ~ 1e-4, the dot products cancel) and lies IN FRONT OF the distance at which the ray enters that
 * triangle's own padded box.  A walk that skips boxes entered beyond its best hit would report such a record or not depending
 * on the order in which it meets the leaves (packet shape, shard count, dispatch history).  The fast modes therefore do not
 * count a computed t that is smaller than the slab test's entry distance of the triangle's own box: their record is the
 * nearest hit that is not in front of its own box, the same whatever the order (and whatever the number of GPUs), and equal
 * to the reference's wherever the reference's winner is not such a t.  The CPU oracle carries the rule as an option
 * (orc_trace_primary_rule) and the fast modes are tested equal to THAT frame bit for bit; an application that wants the
 * reference's artefacts too has LBVH_TRACE_REFERENCE.  Frequency: one pixel in the order of 10^9 randomised rays
 * (tools/fuzz_parity.py), none of the 2 073 600 of the benchmark's frame.  The secondary-ray calls (lbvh_trace_rays, the path
 * tracer) use the same rule. */

/* Optional per-launch traversal statistics (sums over all rays of the launch), in the
 * reference's visit semantics for LBVH_TRACE_REFERENCE: P nodes popped, B internal boxes hit,
 * L leaf-AABB tests, T triangle tests.  Used for the algorithmic-bytes figure.
 * LBVH_TRACE_FAST walks one 8x8-pixel packet per wave: there `pops` = 64-byte node fetches and
 * `leaf_tests` = triangle-line fetches per PACKET (each shared by the packet's 64 rays),
 * `box_hits` and `tri_tests` stay per ray. */
typedef struct lbvh_trace_stats {
    uint64_t pops;
    uint64_t box_hits;
    uint64_t leaf_tests;
    uint64_t tri_tests;
    uint64_t hits;
} lbvh_trace_stats;

typedef struct lbvh_context lbvh_context;

/* ---- library / context ---------------------------------------------------------------------- */

/* ABI version of the loaded library (== LBVH_ABI_VERSION of the header it was built from). */
int32_t lbvh_abi_version(void);

/* Number of visible HIP devices; does not create a context.  Returns <0 on error. */
int32_t lbvh_device_count(void);

/* Create a context on HIP device `device_id` with its own non-blocking stream.
 * Replaces: the implicit Unity graphics device + IShaderContainer kernel registry
 * (Sc/ShaderContainer.cs:6-40, FindKernel calls Sc/ComputeBufferSorter.cs:64-82,
 * Sc/BVHConstructor.cs:45-46, Sc/RaytracingMeshDrawer.cs:59-60). */
lbvh_status lbvh_create(int32_t device_id, lbvh_context** out_ctx);

/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch's current stream); the context
 * never destroys that stream. */
lbvh_status lbvh_create_on_stream(int32_t device_id, void* hip_stream, lbvh_context** out_ctx);

/* Frees the context's scratch (sort ping-pong, histograms, refit scratch) and its stream.
 * Replaces the Dispose chains (Sc/ComputeBufferSorter.cs:274-281, Sc/BVHConstructor.cs:71-74). */
lbvh_status lbvh_destroy(lbvh_context* ctx);

/* Text of the last error on this context ("" if none).  ctx may be NULL for creation errors. */
const char* lbvh_last_error(const lbvh_context* ctx);

/* Block until everything enqueued on the context's stream has finished. */
lbvh_status lbvh_sync(lbvh_context* ctx);

/* ---- buffers: DataBuffer<T> / ComputeBuffer (Sc/DataBuffer.cs) ------------------------------ */

/* new ComputeBuffer(count, stride, Structured) — Sc/DataBuffer.cs:25-30. Contents undefined. */
lbvh_status lbvh_buffer_alloc(lbvh_context* ctx, size_t count, size_t stride, void** out_d_ptr);
/* ComputeBuffer.Release — Sc/DataBuffer.cs:72-75 */
lbvh_status lbvh_buffer_free(lbvh_context* ctx, void* d_ptr);
/* DataBuffer(size, initialValue) for word-patterned values — Sc/DataBuffer.cs:14-23, used with
 * 0xFFFFFFFF for keys/indices/NullLeaf nodes (Sc/MeshBufferContainer.cs:108-109,114-115) and 0
 * for the refit flags (Sc/BVHConstructor.cs:41).  Stream-ordered. */
lbvh_status lbvh_buffer_fill_u32(lbvh_context* ctx, void* d_ptr, uint32_t value, size_t n_words);
/* ComputeBuffer.SetData — Sc/DataBuffer.cs:56-60.  Stream-ordered, host buffer is consumed
 * before the call returns. */
lbvh_status lbvh_buffer_upload(lbvh_context* ctx, void* d_dst, const void* h_src, size_t bytes);
/* ComputeBuffer.GetData — Sc/DataBuffer.cs:50-54.  Blocking. */
lbvh_status lbvh_buffer_download(lbvh_context* ctx, void* h_dst, const void* d_src, size_t bytes);

/* ---- stage a-1: Morton codes + per-triangle AABBs ------------------------------------------ */

/* Replaces the CPU loop of MeshBufferContainer's constructor (Sc/MeshBufferContainer.cs:123-146
 * with GetCentroidAndAABB :52-71, NormalizeCentroid :73-83, Morton3D/ExpandBits :32-50).
 * For i < n: d_keys[i] = Morton3D of the padded-AABB centre normalised to the scene box
 * [box_min, box_max] (the reference hard-wires +-125, :9-15), d_indices[i] = i,
 * d_aabb[i] = {min3-0.001, 0, max3+0.001, 0}.  Strict fp32, no FMA contraction.
 * For n <= i < capacity: d_keys[i] = d_indices[i] = 0xFFFFFFFF (the fill of :108-109);
 * d_aabb is left untouched there.  Requires capacity >= n. */
lbvh_status lbvh_morton_aabb(lbvh_context* ctx, const lbvh_triangle* d_triangles, uint32_t n,
                             uint32_t capacity, const float h_box_min[3], const float h_box_max[3],
                             uint32_t* d_keys, uint32_t* d_indices, lbvh_aabb* d_aabb);

/* ---- stage a-2..a-5: radix sort of (key, value) pairs --------------------------------------- */

/* Replaces ComputeBufferSorter<uint,uint>.Sort() (Sc/ComputeBufferSorter.cs:100-126) and its five
 * kernels (Sh/Sorting/LocalRadixSort.compute:53-134, Sh/Sorting/Scan.compute:15-96,
 * Sh/Sorting/GlobalRadixSort.compute:20-40): sorts `count` (key,value) pairs in place, ascending
 * by key, STABLE (equal keys keep their input order) — the unique result of the reference's
 * 4-pass LSD radix sort.  The reference always sorts its whole padded capacity; pass the capacity
 * as `count` to reproduce that (0xFFFFFFFF pads end up last).  Any count >= 0 is accepted.
 * LATENCY (the result never depends on it): for 2^15 <= count < 2^21 a context whose last THREE sorts of this size class had
 * their keys spread over the 12-bit key prefixes (no more than ~12 K pairs under one prefix) takes a two-level form — 49 us
 * instead of 62 at 1 M pairs — whose one weak case is the FIRST input after such a streak that puts more than 16 384 pairs
 * under one prefix: that one call is sorted by a single workgroup per oversized bucket (2 M pairs under one prefix: 13 ms
 * against 0.12 ms; all keys equal: 1.2 ms; profiles/r6/n_sort_cliff.txt), the following calls take the four passes again.
 * An input that alternates between the two kinds never leaves the four passes; a sort captured into a hipGraph (the build
 * chain of lbvh_build_scene when it is replayed) always takes them. */
lbvh_status lbvh_sort_pairs(lbvh_context* ctx, uint32_t* d_keys, uint32_t* d_values,
                            uint32_t count);

/* ---- cfg4: building blocks of the multi-GPU (key-range sharded) sort, SURVEY 8(e) ------------- *
 * The reference has no multi-GPU path; these are the local kernels of the sharded sort that
 * unitysimpleraytracing_amd/sharded_sort.py drives (one process per GPU): every rank sorts its own
 * block, the ranks agree on W-1 splitter keys by all-reducing (RCCL) MSD digit histograms,
 * exchange key ranges with one all-to-all and sort what they received.  The concatenation over
 * ranks is bit-identical to lbvh_sort_pairs over the whole array (stability: ties never straddle
 * ranks and arrive in source-rank order). */

/* d_hist[p * 256 + d] = number of keys k among d_keys[0..count) with ((k >> shift) & 255) == d and,
 * when prefix_shift < 32, (k >> prefix_shift) == h_prefixes[p].  prefix_shift == 32 selects every
 * key (then n_prefixes must be 1 and h_prefixes may be NULL).  n_prefixes <= 16.  d_hist is
 * overwritten.  Keys need not be sorted. */
lbvh_status lbvh_key_histogram(lbvh_context* ctx, const uint32_t* d_keys, uint32_t count,
                               const uint32_t* h_prefixes, uint32_t n_prefixes, uint32_t prefix_shift,
                               uint32_t shift, uint32_t* d_hist);

/* d_positions[j] = the first i in [0, count] with d_sorted_keys[i] >= h_probes[j] (count if none);
 * n_probes <= 64. */
lbvh_status lbvh_lower_bound(lbvh_context* ctx, const uint32_t* d_sorted_keys, uint32_t count,
                             const uint32_t* h_probes, uint32_t n_probes, uint32_t* d_positions);

/* The same two with the prefixes / probes read from DEVICE memory (u32 arrays of n entries): the sharded sort keeps
 * its splitter search on the device between the RCCL all-reduces — digit histogram, all-reduce, digit selection
 * (a few torch operations on the [W-1][256] table), next histogram — with no host round trip per round. */
lbvh_status lbvh_key_histogram_device(lbvh_context* ctx, const uint32_t* d_keys, uint32_t count,
                                      const uint32_t* d_prefixes, uint32_t n_prefixes, uint32_t prefix_shift,
                                      uint32_t shift, uint32_t* d_hist);
lbvh_status lbvh_lower_bound_device(lbvh_context* ctx, const uint32_t* d_sorted_keys, uint32_t count,
                                    const uint32_t* d_probes, uint32_t n_probes, uint32_t* d_positions);

/* ---- stage a-6: DistributeKeys ---------------------------------------------------------------- */

/* Replaces MeshBufferContainer.DistributeKeys (Sc/MeshBufferContainer.cs:154-169), a serial CPU
 * pass between two full-buffer transfers in the reference: on the first n SORTED keys,
 * new[0] = 0, new[i] = new[i-1] + max(old[i] - old[i-1], 1)  (u32 arithmetic), in place. */
lbvh_status lbvh_distribute_keys(lbvh_context* ctx, uint32_t* d_keys, uint32_t n);

/* ---- stage a-7: Karras LBVH topology ---------------------------------------------------------- */

/* Replaces BVHConstructor.ConstructTree -> kernel TreeConstructor
 * (Sc/BVHConstructor.cs:61-64, Sh/BVH/BVH.compute:18-149).  d_sorted_keys[0..n) must be strictly
 * increasing (true after lbvh_distribute_keys).  Writes internal nodes [0, n-1) and leaf nodes
 * [0, n); words the reference never writes (root.parent) are left as they are — fill the buffers
 * with 0xFFFFFFFF first, as the reference does.  n >= 2 (the reference underflows for n < 2,
 * BVH.compute:101). */
lbvh_status lbvh_build_tree(lbvh_context* ctx, uint32_t n, const uint32_t* d_sorted_keys,
                            lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf);

/* ---- stage a-8: bottom-up AABB refit ------------------------------------------------------------ */

/* Replaces BVHConstructor.ConstructBVH -> kernel BVHConstructor
 * (Sc/BVHConstructor.cs:66-69, Sh/BVH/BVH.compute:152-220): d_bvh[i] = union of the leaf AABBs under
 * internal node i (min/max are exact, so the result equals the reference's arrival-order merge bit for
 * bit).  The reference's per-node arrival flags (atomicsData, Sc/BVHConstructor.cs:41, zeroed once) have
 * no counterpart: the context owns the scratch of the refit and every call is self-contained, so the
 * tree can be rebuilt per frame.  d_internal / d_leaf must hold a tree in ConstructTree's numbering
 * (left child index = split, right = split + 1, root = node 0); the root's parent word is not read.
 * d_triangle_aabb is in ORIGINAL triangle order and is gathered through d_sorted_indices, as in the
 * reference (:203,:212). */
lbvh_status lbvh_refit(lbvh_context* ctx, uint32_t n, const lbvh_internal_node* d_internal,
                       const lbvh_leaf_node* d_leaf, const lbvh_aabb* d_triangle_aabb,
                       const uint32_t* d_sorted_indices, lbvh_aabb* d_bvh);

/* ---- the whole build chain in one call ----------------------------------------------------------- */

#define LBVH_BUILD_FAST_SCENE  1u   /* also build the derived traversal scene (= lbvh_build_fast_scene)        */
#define LBVH_BUILD_RESET_NODES 2u   /* d_internal / d_leaf come out as after a refill with 0xFFFFFFFF           *
                                     * (capacity slots: the slots past the tree and the root's parent word are  *
                                     * written, every other word is the tree's)                                 */

/* RaytracingMeshDrawer.Awake()'s build chain (Sc/RaytracingMeshDrawer.cs:34-51) = lbvh_morton_aabb ->
 * lbvh_sort_pairs(capacity) -> lbvh_distribute_keys -> lbvh_build_tree -> lbvh_refit (+ lbvh_build_fast_scene
 * with LBVH_BUILD_FAST_SCENE), with identical results.  After the sort the reference's arrays and the derived
 * traversal scene are two independent chains of short latency-bound kernels; this call runs them side by side —
 * sharing three launches on the context's stream (up to 2 M triangles), or on the stream and an internal side
 * stream joined before control returns to the stream — so later calls on the context see both.  For per-frame
 * rebuilds of dynamic scenes. */
lbvh_status lbvh_build_scene(lbvh_context* ctx, const lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                             const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys,
                             uint32_t* d_indices, lbvh_aabb* d_aabb, lbvh_internal_node* d_internal,
                             lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh, uint32_t flags);

/* ---- stage a-9: primary-ray traversal ------------------------------------------------------------ */

/* The scene buffers RaytracingMeshDrawer binds to the Raytracing kernel
 * (Sc/RaytracingMeshDrawer.cs:65-70).  All device pointers. */
typedef struct lbvh_scene {
    uint32_t                  n;                 /* triangle count                                 */
    const uint32_t*           sorted_indices;    /* sortedTriangleIndices                          */
    const lbvh_aabb*          triangle_aabb;     /* triangleAABB (original order)                  */
    const lbvh_internal_node* internal_nodes;    /* internalNodes                                  */
    const lbvh_leaf_node*     leaf_nodes;        /* leafNodes                                      */
    const lbvh_aabb*          bvh;               /* bvhData                                        */
    const lbvh_triangle*      triangles;         /* triangleData (original order)                  */
} lbvh_scene;

/* LBVH_TRACE_FAST and the secondary-ray calls work from a derived traversal scene the context caches.  The cache is
 * keyed: it answers only for the scene it was built from — same triangles / sorted_indices / triangle_aabb pointers and
 * n — and only while no library call has written into those buffers since (lbvh_morton_aabb, lbvh_sort_pairs,
 * lbvh_buffer_upload / fill_u32, lbvh_animate, lbvh_build_scene without LBVH_BUILD_FAST_SCENE ...).  Otherwise the
 * trace returns LBVH_ERR_INVALID_ARG ("stale") instead of hits from old geometry: the reference's Dispatch has no hidden
 * state (Sc/RaytracingMeshDrawer.cs:65-70).  Writes the library cannot see (the caller's own kernels on those buffers)
 * are the caller's to follow with lbvh_build_fast_scene.
 *
 * Build the derived traversal structure used by LBVH_TRACE_FAST for `scene`: its own "traversal
 * tree" over the scene's SORTED triangle order (Karras topology over minimally perturbed Morton
 * keys k'_i = i + max_{j<=i}(k_j - j) instead of DistributeKeys' shifted ones — tighter boxes — and
 * its own refit), flattened to fused 64-B nodes (both child boxes + child references) plus the sorted
 * triangles as 64-B lines (first vertex, two edge vectors, original index) in the same array.  At most 2^30 - 1
 * triangles.  h_box_min/max = the scene box given to
 * lbvh_morton_aabb (the raw Morton code of a sorted position is recomputed from its triangle AABB).
 * Owned by the context, rebuilt on each call; call it after lbvh_sort_pairs (+ lbvh_morton_aabb) and
 * before the first LBVH_TRACE_FAST launch.  The scene's internalNodes / leafNodes / bvhData are not
 * read.  No reference counterpart: hit results are those of the reference arrays (every leaf keeps
 * the slab test of its own AABB), only the order and number of node visits differ. */
lbvh_status lbvh_build_fast_scene(lbvh_context* ctx, const lbvh_scene* h_scene, const float h_box_min[3],
                                  const float h_box_max[3]);

/* Replaces RaytracingMeshDrawer.Update's Dispatch of kernel Raytracing
 * (Sc/RaytracingMeshDrawer.cs:76-84, Sh/Raytracing/Raytracing.compute:105-185) up to and
 * including the traversal loop: for every pixel (x, y) with x0 <= x < x1, y0 <= y < y1 generates
 * the camera ray (:108-126), traverses (:133-176) and writes the RaycastResult to
 * d_hits[(y - y0) * (x1 - x0) + (x - x0)].  The full frame is (0, 0, W, H); rectangles are one way to
 * shard rays across GPUs (lbvh_trace_primary_shard below is the other).  Pixels outside the screen are never computed (the reference
 * over-dispatches and relies on D3D dropping out-of-bounds writes, RaytracingMeshDrawer.cs:83).
 * d_stats may be NULL; if not it receives the launch's lbvh_trace_stats (device memory). */
lbvh_status lbvh_trace_primary(lbvh_context* ctx, const lbvh_camera* h_camera,
                               int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                               const lbvh_scene* h_scene, int32_t mode,
                               lbvh_hit* d_hits, lbvh_trace_stats* d_stats);

/* Ray sharding across GPUs in ONE launch per GPU: traces the pixels of shard `shard_index` of
 * `shard_count` of the FULL frame — every shard_count-th group of 8 adjacent 8x8-pixel tiles (a 64x8-
 * pixel strip), so every shard samples the whole frame evenly — and writes them at their full-frame
 * positions d_hits[y * W + x]; pixels of other shards are not touched.  d_hits holds W*H records on
 * every GPU.  The BVH is replicated; no collective is involved.  The union over all shards equals
 * lbvh_trace_primary(0, 0, W, H). */
lbvh_status lbvh_trace_primary_shard(lbvh_context* ctx, const lbvh_camera* h_camera, uint32_t shard_index,
                                     uint32_t shard_count, const lbvh_scene* h_scene, int32_t mode,
                                     lbvh_hit* d_hits, lbvh_trace_stats* d_stats);

/* ---- one frame from N GPUs (BASELINE configs[2]; SURVEY 8(e): "hit records gathered to GPU 0") ----------------------- *
 * The reference renders ONE image per Update() (Sc/RaytracingMeshDrawer.cs:76-89).  With the rays sharded over N GPUs
 * (lbvh_trace_primary_shard, BVH replicated) the frame is whole only once every share's records sit in one full-frame
 * buffer on the GPU that shades / displays it.  Two ways to get them there, no collective on either:
 *   peer-mapped stores   d_hits of lbvh_trace_primary_shard may be memory of ANOTHER GPU of the node (the frame buffer of
 *                        GPU 0): after lbvh_peer_enable the trace kernel's stores travel over xGMI as the tiles finish — no
 *                        second pass, the transfer overlaps the walk.  One process: the owner waits with lbvh_event_wait on
 *                        the tracing contexts' events.  One process per GPU: the buffer crosses with lbvh_ipc_export /
 *                        lbvh_ipc_import and completion with lbvh_frame_signal / lbvh_frame_wait (flag words in the
 *                        owner's memory, written over xGMI, waited for on the device — no host round trip per frame);
 *   packed shares        lbvh_trace_primary_shard_packed writes a share contiguously (work-item order, 64 records per
 *                        8x8 tile) so that any transport can move it as one block (hipMemcpyPeerAsync, an RCCL
 *                        send / recv); lbvh_frame_unpack on the owner puts every share's records at their pixels. */

/* Kernels of this context may read and write memory that lives on `peer_device` (hipDeviceEnablePeerAccess; a no-op for
 * the context's own device or when already enabled).  LBVH_ERR_HIP when the two GPUs cannot reach each other. */
lbvh_status lbvh_peer_enable(lbvh_context* ctx, int32_t peer_device);

/* An event for ORDERING work between contexts (lbvh_event_create's are for timing: they skip the system-scope release
 * that makes a GPU's stores visible to another one).  Created with hipEventDisableTiming | hipEventReleaseToSystem: its record
 * IS a system-scope release whatever the runtime's default for plain events is.  Recorded with lbvh_event_record, destroyed
 * with lbvh_event_destroy; lbvh_event_elapsed_ms does not take it. */
lbvh_status lbvh_sync_event_create(lbvh_context* ctx, void** out_event);
/* The context's stream waits (on the device, the host does not block) for `event` — an event of lbvh_sync_event_create
 * recorded with lbvh_event_record on ANY context of this process, e.g. the end of another GPU's share of the frame. */
lbvh_status lbvh_event_wait(lbvh_context* ctx, void* event);

/* Cross-process handle of a buffer of lbvh_buffer_alloc (hipIpcGetMemHandle, 64 bytes, to be sent to the other process by
 * any means) / the same memory mapped into this process and usable as a d_ pointer on this context (after
 * lbvh_peer_enable when it lives on another GPU) / unmapped again.  The exporting process keeps ownership. */
#define LBVH_IPC_HANDLE_BYTES 64
lbvh_status lbvh_ipc_export(lbvh_context* ctx, void* d_ptr, uint8_t h_handle[LBVH_IPC_HANDLE_BYTES]);
lbvh_status lbvh_ipc_import(lbvh_context* ctx, const uint8_t h_handle[LBVH_IPC_HANDLE_BYTES], void** out_d_ptr);
lbvh_status lbvh_ipc_close(lbvh_context* ctx, void* d_ptr);

/* Memory for the completion flags below: n_words zeroed 32-bit words that a RUNNING kernel may poll while another GPU or
 * process stores into them.  Ordinary device memory (lbvh_buffer_alloc: coarse-grained) only promises visibility of another
 * device's stores at kernel boundaries, so a flag polled inside a kernel could be served stale from this GPU's L2 for as long
 * as the kernel runs; these words are allocated uncached (hipExtMallocWithFlags, hipDeviceMallocUncached — what RCCL uses for
 * its own flags): every load and store goes to the memory itself.  Exported / imported / freed like any other buffer
 * (lbvh_ipc_export, lbvh_ipc_import, lbvh_buffer_free). */
lbvh_status lbvh_flags_alloc(lbvh_context* ctx, size_t n_words, uint32_t** out_d_flags);
/* d_flags[slot] := value once everything enqueued on this context so far has finished and its stores are visible system
 * wide (a release store at system scope; d_flags may be another GPU's memory: words of lbvh_flags_alloc).  Values of one slot
 * must grow (frame numbers). */
lbvh_status lbvh_frame_signal(lbvh_context* ctx, uint32_t* d_flags, uint32_t slot, uint32_t value);
/* Work enqueued on this context after the call starts only when d_flags[s] >= value (as signed distance: wrap-around safe)
 * for every s < n_slots — e.g. every other rank has signalled this frame.  The wait runs on the device and is bounded by WALL
 * CLOCK (20 s of the GPU's constant 100 MHz counter — a rank may still be starting up when the wait is enqueued; a poll count
 * would make the bound depend on how fast the polls come back): if a flag has not arrived by then the next lbvh_sync /
 * lbvh_buffer_download returns LBVH_ERR_HIP, the GPU does not hang.  n_slots <= 64.
 * d_flags may be another GPU's memory (polled over xGMI: the ranks that wait for the owner's "frame read" word). */
lbvh_status lbvh_frame_wait(lbvh_context* ctx, const uint32_t* d_flags, uint32_t n_slots, uint32_t value);

/* lbvh_trace_primary_shard with the share written CONTIGUOUSLY: record of lane l (pixel (l & 7, l >> 3) of the tile) of the
 * share's k-th work item at d_packed[k * 64 + l]; work item k is tile ((k / 8) * shard_count + shard_index) * 8 + k % 8 of the
 * full frame in row-major tile order.  lbvh_shard_records gives the number of records a share occupies (whole tiles: lanes
 * outside the screen and tiles past the frame's last one are never written). */
lbvh_status lbvh_trace_primary_shard_packed(lbvh_context* ctx, const lbvh_camera* h_camera, uint32_t shard_index,
                                            uint32_t shard_count, const lbvh_scene* h_scene, int32_t mode,
                                            lbvh_hit* d_packed, lbvh_trace_stats* d_stats);
/* Records of shard `shard_index`'s packed share of a width x height frame (pure function; the same for every call site). */
uint64_t lbvh_shard_records(int32_t width, int32_t height, uint32_t shard_index, uint32_t shard_count);
/* Packed shares -> the full frame: share s (s in [first_shard, first_shard + n_shards)) starts at
 * d_packed[(s - first_shard) * share_stride] (share_stride in records, >= lbvh_shard_records of any of them); its records
 * go to d_frame_hits[y * width + x].  Pixels of other shards are not touched. */
lbvh_status lbvh_frame_unpack(lbvh_context* ctx, const lbvh_hit* d_packed, uint64_t share_stride, uint32_t first_shard,
                              uint32_t n_shards, uint32_t shard_count, int32_t width, int32_t height, lbvh_hit* d_frame_hits);

/* ---- stage a-9, tail: shading -------------------------------------------------------------------- */

/* Replaces the rest of kernel Raytracing's body (Sh/Raytracing/Raytracing.compute:178-184) for `count`
 * RaycastResults in any layout:  t = triangleData[hit.tri] (triangle 0 on a miss, as in the reference),
 * uv = (1-u-v) a_uv + u b_uv + v c_uv, normal likewise (:179-180), lightDir = the SCALAR 0.57735026
 * (normalize(float3(1,1,1)) assigned to a `float`, :181), colour = texture(uv).rgb * max(0.4,
 * dot(lightDir, normal)) (:183), output float4(colour, hit ? 1 : 0) stored as RGBA16F like the
 * reference's render target (Sc/RaytracingMeshDrawer.cs:56).
 * Texture: d_texture_rgba8 = tex_h rows of tex_w RGBA8 texels, row 0 at v = 0 (Unity's convention), no
 * sRGB decode (the project is in Gamma colour space, ProjectSettings.asset:50), sampled like
 * SampleLevel(linearClampSampler, uv, 0): bilinear on texel centres in fp32, clamp addressing, mip 0.
 * d_rgba16f receives count x 4 IEEE half floats. */
lbvh_status lbvh_shade(lbvh_context* ctx, const lbvh_hit* d_hits, size_t count, const lbvh_triangle* d_triangles,
                       const uint8_t* d_texture_rgba8, int32_t tex_w, int32_t tex_h, uint16_t* d_rgba16f);

/* Replaces the full-screen pass of Hidden/ImageComposer (Sh/ImageComposer.shader:44-52, driven by
 * RaytracingMeshDrawer.OnRenderImage, Sc/RaytracingMeshDrawer.cs:86-90): the traced image is laid over the camera's
 * own rendering, one texel over the same pixel:  out.rgb = lerp(background.rgb, object.rgb, object.a)
 * = background + object.a * (object - background)  in fp32, out.a = 1.  All three images: count x 4 IEEE halfs
 * (RGBA16F, the format of lbvh_shade's output).  d_out may alias d_background. */
lbvh_status lbvh_compose(lbvh_context* ctx, const uint16_t* d_background_rgba16f, const uint16_t* d_object_rgba16f,
                         size_t count, uint16_t* d_out_rgba16f);

/* ---- dynamic scenes and secondary rays (extension; no reference counterpart) ----
 * The reference traces primary rays of a static mesh only (Sh/Raytracing/Raytracing.compute has no
 * secondary rays and no RNG; Sc/BVHConstructor.cs:41 zeroes the refit flags once, so it cannot even
 * rebuild).  BASELINE configs[4] asks for a per-frame rebuild + 4-bounce 1-spp path trace; the pieces
 * below provide it.  Parity for them is against this repo's own CPU restatement (oracle/), bit for bit:
 * everything is strict fp32, trig-free, and driven by a counter-based RNG. */

/* Rigid per-body animation: triangle i of the rest pose belongs to body d_body[i]; its three positions
 * and normals are rotated about the Y axis through that body's centre h/d_centres[body] by the angle
 * whose cosine / sine the HOST passes (no device trig), uv copied.  d_out may not alias d_rest. */
lbvh_status lbvh_animate(lbvh_context* ctx, const lbvh_triangle* d_rest, uint32_t n, const uint32_t* d_body,
                         const float* d_centres /* n_bodies x 4 floats (xyz, pad) */, float cos_angle, float sin_angle,
                         lbvh_triangle* d_out);

/* lbvh_animate followed by lbvh_build_scene as one call, with identical results: the moved triangles go to d_triangles AND
 * straight into the Morton / AABB stage (one kernel: as two, the 128-byte records were written and read back for the 36 bytes
 * of their positions).  The per-frame chain of a dynamic scene (BASELINE configs[4]); every other argument as lbvh_build_scene. */
lbvh_status lbvh_animate_build_scene(lbvh_context* ctx, const lbvh_triangle* d_rest, const uint32_t* d_body, const float* d_centres,
                                     float cos_angle, float sin_angle, lbvh_triangle* d_triangles, uint32_t n, uint32_t capacity,
                                     const float h_box_min[3], const float h_box_max[3], uint32_t* d_keys, uint32_t* d_indices,
                                     lbvh_aabb* d_aabb, lbvh_internal_node* d_internal, lbvh_leaf_node* d_leaf, lbvh_aabb* d_bvh,
                                     uint32_t flags);

/* One path vertex per pixel: 64 bytes. */
typedef struct lbvh_path_state {
    float origin[3];     uint32_t alive;     /* 1 while the path continues                      */
    float dir[3];        float    pad0;      /* unit direction of the ray to trace next          */
    float throughput[3]; float    pad1;
    float radiance[3];   float    alpha;     /* alpha = 1 if the primary ray hit, as Raytracing.compute:184 */
} lbvh_path_state;

/* Closest hit for `count` arbitrary rays taken from the path states (origin, dir; dead paths are skipped and
 * get a miss record): one ray per lane over the derived traversal scene (lbvh_build_fast_scene) in its four-wide
 * form — every node of that tree with its largest children opened once or twice: up to four child boxes per 128-byte
 * line, half the steps of the binary walk; made by the first call after a rebuild (collapse_wide_kernel, 0.06 ms at
 * 1 M triangles) — nearest child first, t-pruned, per-lane stack of 128 entries (three siblings can wait per level;
 * the reference's binary walk has 64, Raytracing.compute:113; the first 16 in LDS, deeper ones in device memory).
 * Accept rule = the reference's (own-AABB slab test, Moeller-Trumbore, t < best) plus t > t_min, which secondary
 * rays need to leave their surface and the reference lacks (Raytracing.compute:70); two triangles hit at the same t:
 * the lower triangle index, whatever order the walk meets them in; a computed t in front of its own triangle's box
 * does not count (both as LBVH_TRACE_FAST: see the note at the traversal flavours). */
lbvh_status lbvh_trace_rays(lbvh_context* ctx, const lbvh_path_state* d_states, size_t count, float t_min,
                            const lbvh_scene* h_scene, lbvh_hit* d_hits);

/* Camera rays into path states (origin/dir as Raytracing.compute:108-126, throughput 1, radiance 0, alive). */
lbvh_status lbvh_path_begin(lbvh_context* ctx, const lbvh_camera* h_camera, lbvh_path_state* d_states);

/* One bounce for every live path given its hit record: a miss adds throughput * sky(dir) and ends the path
 * (sky = (1 - s) * (1,1,1) + s * (0.5,0.7,1), s = 0.5 * (dir.y + 1)); a hit multiplies the throughput by
 * `albedo`, moves the origin to the hit point and draws a cosine-weighted direction about the geometric
 * normal (flipped toward the incoming ray): normalize(n + p) with p uniform on the unit sphere by
 * Marsaglia's rejection method, random numbers = PCG hash of (seed, path index, bounce, draw).  `bounce` = 0
 * for the primary hit (it also sets alpha). */
lbvh_status lbvh_path_scatter(lbvh_context* ctx, const lbvh_scene* h_scene, const lbvh_hit* d_hits, size_t count,
                              uint32_t bounce, uint32_t seed, float albedo, lbvh_path_state* d_states);

/* lbvh_path_scatter for bounce `bounce` followed by lbvh_trace_rays for the next segment of the paths that go on,
 * as one call: the scatter kernel itself lists those paths, so no pass over all path states is needed before the
 * trace.  d_hits holds the hit records of the segment just traced on entry and those of the next segment on
 * return.  The record of a path that ends in this call (it ends on a miss) becomes {t = MAX_FLOAT, triangle =
 * 0xFFFFFFFF, 0, 0} — still a miss to every reader; a later lbvh_path_bounce (bounce > 0) on the same buffers recognises
 * it and skips the finished path without reading its 64-byte state.  At bounce 0 every record is the caller's: one that
 * was pre-filled with 0xFFFFFFFF words and never traced is an ordinary miss (sky term, path ends), as in
 * lbvh_path_scatter.  States, radiance and image: the same as the two calls.
 * CROSS-CALL STATE: a call with bounce >= 1 — and the frame's last lbvh_path_scatter — visits only the paths the previous
 * lbvh_path_bounce on the same d_states / d_hits listed as live (a list kept by the context).  Every library call that writes
 * into those buffers drops the list (lbvh_path_begin, lbvh_trace_rays, a primary trace into any part of d_hits,
 * lbvh_buffer_upload / _fill_u32 / _free), and so does lbvh_trace_forget; then every state is scanned again.  What the library
 * cannot see is a write of the CALLER's own (a kernel or hipMemcpy that revives or ends paths, Russian roulette): between two
 * consecutive bounces of a frame d_states and d_hits must not be written from outside the library — or lbvh_trace_forget must be
 * called after such a write. */
lbvh_status lbvh_path_bounce(lbvh_context* ctx, const lbvh_scene* h_scene, lbvh_path_state* d_states, lbvh_hit* d_hits,
                             size_t count, uint32_t bounce, uint32_t seed, float albedo, float t_min);

/* lbvh_path_begin + lbvh_path_bounce(bounce = 0) as one call: d_hits holds the primary hit records of the camera's
 * W x H frame (lbvh_trace_primary); every pixel's path state is MADE from the camera on the fly — never stored by one
 * kernel to be loaded by the next (132 MB each way at 1080p) — scattered at its hit, and the first secondary segment is
 * traced.  States, hit records and the image: identical to the two calls. */
lbvh_status lbvh_path_first_bounce(lbvh_context* ctx, const lbvh_camera* h_camera, const lbvh_scene* h_scene, lbvh_path_state* d_states,
                                   lbvh_hit* d_hits, uint32_t seed, float albedo, float t_min);

/* radiance (+ alpha) of the path states as RGBA16F, the reference's render-target format. */
lbvh_status lbvh_path_resolve(lbvh_context* ctx, const lbvh_path_state* d_states, size_t count, uint16_t* d_rgba16f);

/* LBVH_TRACE_FAST dispatches a frame's tiles in the order of their step counts in the PREVIOUS trace of the same frame
 * layout (a scheduling hint kept by the context; any order gives the same hits).  When the camera differs from that
 * trace's, a tile takes the count of the place it came from: the ray through its centre, at the distance of the scene
 * box's centre, projected with the previous camera (exact for a turn of the camera), widened by one tile.  This call
 * drops the history: the next trace runs as a first frame does (row-major).  For measuring cold frames.  It also drops the path
 * tracer's live-path list (lbvh_path_bounce): the next bounce scans every path state. */
lbvh_status lbvh_trace_forget(lbvh_context* ctx);

/* Multi-GPU frames (one context per GPU, each tracing its lbvh_trace_primary_shard share): the dispatch hint above comes
 * from the context's OWN last trace, and under a moving camera the place a tile came from mostly belongs to another
 * rank.  lbvh_trace_costs_export writes this context's per-tile step counts of its last LBVH_TRACE_FAST trace into a
 * full-frame array (one u32 per 8x8-pixel tile, row-major, ceil(W/8) x ceil(H/8); tiles of other shards are left as they
 * are — hand in a zeroed array); the ranks merge their arrays (an all-reduce MAX or SUM of 130 KB at 1080p, e.g. while
 * the next rebuild runs) and give the result back with lbvh_trace_costs_import (copied).  The next trace with a DIFFERENT
 * camera takes its tiles' costs from there — once: a map is a hint for the frame that follows it, not for later ones.  A hint only: hits do
 * not depend on it.  Both calls are asynchronous on the context's stream. */
lbvh_status lbvh_trace_costs_export(lbvh_context* ctx, uint32_t* d_frame_costs, uint32_t tiles_x, uint32_t tiles_y);
lbvh_status lbvh_trace_costs_import(lbvh_context* ctx, const uint32_t* d_frame_costs, uint32_t tiles_x, uint32_t tiles_y);

/* ---- measurement helpers (HIP events on the context's stream) --------------------------------- */

lbvh_status lbvh_event_create(lbvh_context* ctx, void** out_event);
lbvh_status lbvh_event_destroy(lbvh_context* ctx, void* event);
lbvh_status lbvh_event_record(lbvh_context* ctx, void* event);
/* Waits for `stop`, then returns the elapsed milliseconds between the two recorded events. */
lbvh_status lbvh_event_elapsed_ms(lbvh_context* ctx, void* start, void* stop, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* LBVH_H */
