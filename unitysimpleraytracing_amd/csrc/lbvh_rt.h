// lbvh_rt.h — ray generation and the reference's intersection arithmetic, shared by the traversal
// kernels (lbvh_trace.hip, lbvh_path.hip).  Strict fp32 in the reference's operation order
// (Assets/_Shaders/Raytracing/Raytracing.compute:37-87, 108-126); the library is built with
// -ffp-contract=off so these round exactly like oracle/lbvh_oracle.c.
#pragma once

#include "lbvh_common.h"

namespace {

struct ray_t {
    float ox, oy, oz;
    float dx, dy, dz;
    float ix, iy, iz;
};

// Raytracing.compute:108-126; same expression order as oracle/lbvh_oracle.c orc_make_ray
__device__ __forceinline__ ray_t make_ray(const lbvh_camera& cam, uint32_t px, uint32_t py)
{
    const float near = cam.near_plane;
    const float fov = cam.camera_fov;
    const float height = 2.0f * near * fov;
    const float width = (float)cam.screen_width * height / (float)cam.screen_height;
    const float d0 = -width / 2.0f + width / (float)cam.screen_width * ((float)px + 0.5f);
    const float d1 = -height / 2.0f + height / (float)cam.screen_height * ((float)py + 0.5f);
    const float d2 = -near;
    const float* m = cam.camera_to_world;
    float o[3], w[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        o[r] = ((m[4 * r + 0] * 0.0f + m[4 * r + 1] * 0.0f) + m[4 * r + 2] * 0.0f) + m[4 * r + 3] * 1.0f;
        w[r] = ((m[4 * r + 0] * d0 + m[4 * r + 1] * d1) + m[4 * r + 2] * d2) + m[4 * r + 3] * 0.0f;
    }
    const float len = sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    ray_t ray;
    ray.ox = o[0]; ray.oy = o[1]; ray.oz = o[2];
    ray.dx = w[0] / len; ray.dy = w[1] / len; ray.dz = w[2] / len;
    ray.ix = 1.0f / ray.dx; ray.iy = 1.0f / ray.dy; ray.iz = 1.0f / ray.dz;
    return ray;
}

// RayBoxIntersection, Raytracing.compute:75-87.  Returns the hit predicate; tmin_out = entry t.
__device__ __forceinline__ bool ray_box(const float4 bmin, const float4 bmax, const ray_t& r, float& tmin_out)
{
    const float t1x = (bmin.x - r.ox) * r.ix, t2x = (bmax.x - r.ox) * r.ix;
    const float t1y = (bmin.y - r.oy) * r.iy, t2y = (bmax.y - r.oy) * r.iy;
    const float t1z = (bmin.z - r.oz) * r.iz, t2z = (bmax.z - r.oz) * r.iz;
    const float tmin = fmaxf(fminf(t1x, t2x), fmaxf(fminf(t1y, t2y), fminf(t1z, t2z)));
    const float tmax = fminf(fmaxf(t1x, t2x), fminf(fmaxf(t1y, t2y), fmaxf(t1z, t2z)));
    tmin_out = tmin;
    return tmax > tmin && tmax > 0.0f;
}

// The same test when the caller has already ordered the planes by the ray's direction signs (near = the plane the
// ray meets first on each axis): with a finite non-zero inverse direction, (near - o) * i <= (far - o) * i holds
// exactly (fp subtraction and multiplication by one factor are monotone), so fminf(t1, t2) IS the near product
// and fmaxf(t1, t2) the far one — six instructions less per box, the same tmin / tmax bit for bit.
__device__ __forceinline__ bool ray_box_ordered(const float nx, const float ny, const float nz, const float fx, const float fy,
                                                const float fz, const ray_t& r, float& tmin_out)
{
    const float tnx = (nx - r.ox) * r.ix, tfx = (fx - r.ox) * r.ix;
    const float tny = (ny - r.oy) * r.iy, tfy = (fy - r.oy) * r.iy;
    const float tnz = (nz - r.oz) * r.iz, tfz = (fz - r.oz) * r.iz;
    const float tmin = fmaxf(tnx, fmaxf(tny, tnz));
    const float tmax = fminf(tfx, fminf(tfy, tfz));
    tmin_out = tmin;
    return tmax > tmin && tmax > 0.0f;
}

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}

// RayTriangleIntersection, Raytracing.compute:37-73, from the first vertex and the two edge vectors
// e1 = v1 - v0, e2 = v2 - v0 (:41-42).  Returns distance (LBVH_MAX_FLOAT = miss).
__device__ __forceinline__ float ray_triangle_edges(const ray_t& r, const float4 v0, const float e1x, const float e1y, const float e1z,
                                                    const float e2x, const float e2y, const float e2z, float& u_out, float& v_out)
{
    // pvec = cross(dir, e2)
    const float px = r.dy * e2z - r.dz * e2y;
    const float py = r.dz * e2x - r.dx * e2z;
    const float pz = r.dx * e2y - r.dy * e2x;
    const float det = dot3(e1x, e1y, e1z, px, py, pz);
    if (det < 1e-8f && det > -1e-8f) return LBVH_MAX_FLOAT;
    const float inv_det = 1.0f / det;
    const float tx = r.ox - v0.x, ty = r.oy - v0.y, tz = r.oz - v0.z;
    const float u = dot3(tx, ty, tz, px, py, pz) * inv_det;
    if (u < 0.0f || u > 1.0f) return LBVH_MAX_FLOAT;
    // qvec = cross(tvec, e1)
    const float qx = ty * e1z - tz * e1y;
    const float qy = tz * e1x - tx * e1z;
    const float qz = tx * e1y - ty * e1x;
    const float v = dot3(r.dx, r.dy, r.dz, qx, qy, qz) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return LBVH_MAX_FLOAT;
    u_out = u;
    v_out = v;
    return dot3(e2x, e2y, e2z, qx, qy, qz) * inv_det;
}

__device__ __forceinline__ float ray_triangle(const ray_t& r, const float4 v0, const float4 v1,
                                              const float4 v2, float& u_out, float& v_out)
{
    return ray_triangle_edges(r, v0, v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, u_out, v_out);
}

// a sorted-triangle line of the derived scene (lbvh_fast_tri) as the three float4 ray_fast_triangle takes:
// {v0, original index}, e1, e2
__device__ __forceinline__ void unpack_fast_triangle(const float4* line, float4& t0, float4& t1, float4& t2)
{
    const float4 q0 = line[0], q1 = line[1], q2 = line[2], q3 = line[3];
    t0 = q0;
    t1 = make_float4(q2.x, q2.y, q2.z, 0.0f);
    t2 = make_float4(q1.w, q2.w, q3.w, 0.0f);
}

// the sorted-triangle record of the derived scene: {v0, original index | e1 | e2} — the edges are the same fp32
// differences the reference forms per test, taken once at build time
__device__ __forceinline__ float ray_fast_triangle(const ray_t& r, const float4 t0, const float4 t1, const float4 t2,
                                                   float& u_out, float& v_out)
{
    return ray_triangle_edges(r, t0, t1.x, t1.y, t1.z, t2.x, t2.y, t2.z, u_out, v_out);
}

// The second half of the fast walkers' accept rule (round 5): a computed t that lies BEFORE the distance at which
// the ray enters the triangle's own padded box does not count.  Such a t is noise of the fp32 triangle test on a ray almost
// inside the triangle's plane (det ~ 1e-4, the dot products cancel); the reference, which prunes nothing, reports it whenever
// the ray's line passes that box — a walk that skips boxes entered beyond its best hit would report it or not depending on
// the order in which it happens to meet the leaves (packet shape, shard count, dispatch history).  With this rule the
// result is the nearest hit that is not in front of its own box, whatever the order: entry distances grow from a box to any
// box inside it, so a candidate that counts satisfies t >= entry(leaf) >= entry(any ancestor), and a subtree skipped
// because entry > best (strictly) cannot hold a candidate at or below best.  (entry = the slab test's tmin, the same six
// fp32 products and minima / maxima as RayBoxIntersection; a NaN entry keeps the candidate, as the oracle's compare does.)
#ifdef LBVH_AB_NO_ACCEPT_RULE      // tools/build_variant.sh: what the rule costs (A/B measurement only; never the product)
__device__ __forceinline__ bool hit_counts(float, float) { return true; }
#else
__device__ __forceinline__ bool hit_counts(float t, float box_entry) { return !(t < box_entry); }
#endif

}  // namespace
