/*
 * lbvh_oracle.h — CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY; see the
 * header of lbvh_oracle.c for what may load it and how it is pinned).
 */
#ifndef LBVH_ORACLE_H
#define LBVH_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/lbvh.h"

#ifdef __cplusplus
extern "C" {
#endif

/* OpenMP threads available to the `threads` arguments below (1 if built without OpenMP). */
int orc_num_threads(void);

/* a-1  Assets/_Scripts/MeshBufferContainer.cs:32-83,108-109,123-146 */
void orc_morton_aabb(const lbvh_triangle* tris, uint32_t n, uint32_t capacity,
                     const float box_min[3], const float box_max[3],
                     uint32_t* keys, uint32_t* indices, lbvh_aabb* aabb, int threads);

/* a-2..a-5, semantic: stable ascending sort by key = result of
 * Assets/_Scripts/ComputeBufferSorter.cs:100-126 */
void orc_sort_pairs(uint32_t* keys, uint32_t* values, uint32_t count);

/* a-2..a-5, literal: emulation of LocalRadixSort/PreScan/BlockSum/GlobalScan/GlobalRadixSort with
 * 32-lane waves; count = tiles*1024, tiles a multiple of 128 (reference: 512).  0 = ok. */
int orc_sort_pairs_literal(uint32_t* keys, uint32_t* values, uint32_t tiles);

/* a-6  Assets/_Scripts/MeshBufferContainer.cs:154-169 */
void orc_distribute_keys(uint32_t* keys, uint32_t n);

/* a-7  Assets/_Shaders/BVH/BVH.compute:18-149.  0 = ok, -1 = n < 2, -2 = keys not unique */
int orc_build_tree(uint32_t n, const uint32_t* sorted_keys, lbvh_internal_node* internal,
                   lbvh_leaf_node* leaf, int threads);

/* a-8  Assets/_Shaders/BVH/BVH.compute:152-220 */
int orc_refit(uint32_t n, const lbvh_internal_node* internal, const lbvh_leaf_node* leaf,
              const lbvh_aabb* triangle_aabb, const uint32_t* sorted_indices, lbvh_aabb* bvh);

/* a-9 pieces  Assets/_Shaders/Raytracing/Raytracing.compute:75-87 and :108-126 */
int orc_ray_box(const float bmin[3], const float bmax[3], const float origin[3],
                const float inv_dir[3]);
float orc_ray_triangle(const float orig[3], const float dir[3], const float a[3], const float b[3], const float c[3]);
void orc_make_ray(const lbvh_camera* cam, uint32_t px, uint32_t py, float origin[3], float dir[3],
                  float inv_dir[3]);

/* a-9  Assets/_Shaders/Raytracing/Raytracing.compute:105-176 for the pixels
 * x = x0, x0+x_step, ... < x1; y likewise (steps > 1 subsample the frame for the algorithmic-bytes
 * counters).  `scene` holds HOST pointers here.  hits is row-major over the sampled grid.
 * 0 = ok, -3 = the reference's 64-entry stack would have overflowed. */
int orc_trace_primary(const lbvh_camera* cam, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                      int32_t x_step, int32_t y_step, const lbvh_scene* scene, lbvh_hit* hits,
                      lbvh_trace_stats* stats, int threads);
/* the same loop with the accept rule of the library's fast modes (fast_rule = 1): a computed t in front of its own triangle's box
 * does not count (lbvh_oracle.c: check_triangle) */
int orc_trace_primary_rule(const lbvh_camera* cam, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                           int32_t x_step, int32_t y_step, const lbvh_scene* scene, lbvh_hit* hits,
                           lbvh_trace_stats* stats, int threads, int fast_rule);

/* a-9 tail  Assets/_Shaders/Raytracing/Raytracing.compute:178-184 (see include/lbvh.h lbvh_shade) */
void orc_shade(const lbvh_hit* hits, size_t count, const lbvh_triangle* tris, const uint8_t* tex, int32_t tex_w,
               int32_t tex_h, uint16_t* rgba16f);

/* SURVEY 8(f) rank 3 (extension, no reference counterpart): see include/lbvh.h lbvh_animate, lbvh_path_*,
 * lbvh_trace_rays.  `s` holds HOST pointers to the reference arrays. */
void orc_animate(const lbvh_triangle* rest, uint32_t n, const uint32_t* body, const float* centres, float c, float s,
                 lbvh_triangle* out);
void orc_path_begin(const lbvh_camera* cam, lbvh_path_state* states);
int orc_trace_rays(const lbvh_path_state* states, size_t count, float t_min, const lbvh_scene* s, lbvh_hit* hits,
                   int threads);
void orc_path_scatter(const lbvh_scene* s, const lbvh_hit* hits, size_t count, uint32_t bounce, uint32_t seed,
                      float albedo, lbvh_path_state* states);
void orc_path_resolve(const lbvh_path_state* states, size_t count, uint16_t* rgba16f);

/* OpenMP forms of the serial stages for the CPU baseline (bench.py cpu_baseline): identical results. */
void orc_sort_pairs_mt(uint32_t* keys, uint32_t* values, uint32_t count, int threads);
void orc_distribute_keys_mt(uint32_t* keys, uint32_t n, int threads);
int orc_refit_mt(uint32_t n, const lbvh_internal_node* internal, const lbvh_leaf_node* leaf,
                 const lbvh_aabb* triangle_aabb, const uint32_t* sorted_indices, lbvh_aabb* bvh, int threads);

/* Awake() build chain on the host (Assets/_Scripts/RaytracingMeshDrawer.cs:30-51). */
int orc_build_all(const lbvh_triangle* tris, uint32_t n, uint32_t capacity, const float box_min[3],
                  const float box_max[3], uint32_t* keys, uint32_t* indices, lbvh_aabb* tri_aabb,
                  lbvh_internal_node* internal, lbvh_leaf_node* leaf, lbvh_aabb* bvh, int threads);

#ifdef __cplusplus
}
#endif
#endif
