// lbvh_path.hip — SURVEY 8(f) rank 3: rigid per-body animation, arbitrary-ray traversal and the bounce
// step of a 1-spp path tracer (BASELINE configs[4]).  No reference counterpart: the reference traces
// primary rays of a static mesh only.  Definitions: include/lbvh.h; bit-exact checker: oracle/.
// Strict fp32 (-ffp-contract=off), no device trig, counter-based RNG.
#include <algorithm>
#include <cstdlib>
#include "lbvh_common.h"
#include "lbvh_rt.h"

namespace {

// ---- animation --------------------------------------------------------------------------------------
// One thread per 16-byte quarter-row of a triangle (8 of them per 128-byte record): a wave's load and its store cover 1 KB of
// consecutive bytes.  (One thread per triangle touched 64 different lines with every one of its 8 loads and 8 stores: 74 us per
// million triangles against the 45 us the 256 MB take at the copy rate.)
__global__ __launch_bounds__(256) void animate_kernel(const lbvh_triangle* __restrict__ rest, uint32_t n,
                                                      const uint32_t* __restrict__ body, const float4* __restrict__ centres,
                                                      float c, float s, lbvh_triangle* __restrict__ out)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(g >> 3), k = (uint32_t)g & 7u;
    if (i >= n) return;
    float4 p = reinterpret_cast<const float4*>(rest)[g];
    if (k < 3u) {                                        // positions a, b, c
        const float4 ctr = centres[body[i]];
        const float x = p.x - ctr.x, z = p.z - ctr.z;
        p.x = (c * x + s * z) + ctr.x;                   // rotation about Y through the body centre
        p.z = (c * z - s * x) + ctr.z;
    } else if (k >= 5u) {                                // normals (k = 3, 4: uv, copied)
        const float nx = p.x, nz = p.z;
        p.x = c * nx + s * nz;
        p.z = c * nz - s * nx;
    }
    reinterpret_cast<float4*>(out)[g] = p;
}

// ---- camera rays -> path states ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void path_begin_kernel(lbvh_camera cam, lbvh_path_state* __restrict__ states)
{
    const uint32_t x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= (uint32_t)cam.screen_width || y >= (uint32_t)cam.screen_height) return;
    const ray_t r = make_ray(cam, x, y);
    float4* o = reinterpret_cast<float4*>(&states[(size_t)y * cam.screen_width + x]);
    o[0] = make_float4(r.ox, r.oy, r.oz, __uint_as_float(1u));
    o[1] = make_float4(r.dx, r.dy, r.dz, 0.0f);
    o[2] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    o[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---- arbitrary rays: one ray per lane over the derived traversal scene -----------------------------------
// Secondary rays are incoherent, so the packet walk of lbvh_trace.hip does not apply: every lane walks on its
// own (64-byte fused nodes, near child first, boxes beyond the best hit skipped) with a stack of its own, laid out
// [entry][lane].  One wave per workgroup, no barriers.
// The first kRayStackLds entries of a lane's stack are in LDS (4 KB per wave: all 32 wave slots of a CU fit; with 34
// entries in LDS only 18 did and the four bounces took 1.70 ms instead of 1.46), deeper ones — rare: the fused tree
// is walked near child first — in a per-wave slab of global memory, up to 64 entries in all like the reference's
// stack (Raytracing.compute:113).
constexpr int kRayStackLds = 16;
constexpr int kRayStackDeep = 48;
// the four-wide walk further down: three siblings can wait per level.  (Neither stack can overflow on a tree of this library:
// the derived tree is a radix tree over unique 32-bit keys, at most 32 levels deep — 32 waiting entries for the binary walk,
// 3 x 32 for the four-wide one, whose levels are binary levels or pairs of them.)
constexpr int kWideStackLds = 16;
constexpr int kWideStackDeep = 112;

// Live rays only: alive_rays_kernel writes the miss record of every dead ray and compacts the indices of the live
// ones (after the first bounce more than half of a frame's paths have left the scene).  (Sorting the live rays by direction octant + Morton code of the origin on top of this was
// measured: the sort costs 0.15 ms per bounce and the walk does not get faster.  Measured again in round 5 on the four-wide
// walkers with counters, profiles/r5/c_ray_sort_between_bounces.txt: HBM-side fetch -17 %, L2 misses -15 %, walk time 1.206 ->
// 1.201 ms — the walk waits for its rays' dependent line fetches at full occupancy, not for bandwidth.)
__global__ __launch_bounds__(256) void alive_rays_kernel(const lbvh_path_state* __restrict__ states, size_t count,
                                                         uint32_t* __restrict__ n_alive, uint32_t* __restrict__ list,
                                                         lbvh_hit* __restrict__ hits)
{
    __shared__ uint32_t s_n, s_base;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    bool alive = false;
    if (i < count) {
        alive = reinterpret_cast<const uint32_t*>(&states[i])[3] != 0u;
        if (!alive) reinterpret_cast<float4*>(hits)[i] = make_float4(LBVH_MAX_FLOAT, __uint_as_float(0u), 0.0f, 0.0f);
    }
    const uint64_t m = __ballot(alive);
    uint32_t wave_ofs = 0;
    if (lane_id() == 0 && m) wave_ofs = atomicAdd(&s_n, (uint32_t)__popcll(m));
    wave_ofs = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_ofs);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = __hip_atomic_fetch_add(n_alive, s_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (alive) list[s_base + wave_ofs + mbcnt64(m)] = (uint32_t)i;
}

// One ray per lane; a wave owns a run of consecutive entries of the live-ray list and REFILLS a lane from it as
// soon as that lane's ray is finished, so a wave's run time is the sum of its rays' steps / 64 and not the step
// count of its longest ray (incoherent rays: mean ~100 steps, maxima of several hundred).  The grid is a fixed
// number of waves (every wave slot of the chip once) and the live rays are dealt out evenly: at least 64 per wave.
// Measured per frame of 4 bounces: no refill 3.75 ms; fixed runs of 128 / 256 / 512 rays 1.83 / 2.62 / 4.44 ms
// (long runs leave most of the chip empty).
constexpr uint32_t kRayWaves = 8192;
// ray scratch: [live-ray count (256 B) | indices of the live rays | deep stack slabs of the launch's waves]
static inline uint32_t ray_waves_of(size_t count) { return (uint32_t)std::min<size_t>(kRayWaves, (count + LBVH_WAVE - 1) / LBVH_WAVE); }
// the slab is indexed by blockIdx.x: one launch needs ray_waves_of(count) of them (a 64-ray call: 12 KB, not 96 MB)
static inline size_t deep_bytes(size_t count) { return (size_t)ray_waves_of(count) * std::max(kWideStackDeep, kRayStackDeep) * LBVH_WAVE * 4; }
static inline size_t list_bytes(size_t count) { return (count * 4 + 255) & ~(size_t)255; }
// ray scratch: [two live-ray counters (256 B: words 0 and 16) | live-ray list 0 | live-ray list 1 | deep stack slabs].  Two lists
// take turns (round 5): lbvh_path_bounce b builds the list of the paths that go on FROM the list bounce b - 1 left behind, not
// from a scan over all pixels — after the first bounce most of a frame's paths are dead (870 k, 350 k, 150 k, 60 k live of 2 M).
static inline size_t ray_scratch_bytes_for(size_t count);
static inline uint32_t* ray_counter(lbvh_context* ctx, uint32_t turn) { return (uint32_t*)ctx->ray_scratch + 16u * turn; }
static inline uint32_t* ray_list(lbvh_context* ctx, size_t count, uint32_t turn) { return (uint32_t*)((char*)ctx->ray_scratch + 256 + turn * list_bytes(count)); }
static inline uint32_t* deep_stacks(lbvh_context* ctx, size_t count) { return (uint32_t*)((char*)ctx->ray_scratch + 256 + 2 * list_bytes(count)); }
static inline size_t ray_scratch_bytes_for(size_t count) { return 256 + 2 * list_bytes(count) + deep_bytes(count); }

__global__ __launch_bounds__(64) void trace_rays_kernel(const lbvh_path_state* __restrict__ states, const uint32_t* __restrict__ n_alive,
                                                        const uint32_t* __restrict__ list, float t_min,
                                                        const lbvh_fast_node* __restrict__ nodes,
                                                        const lbvh_fast_tri* __restrict__ tris, lbvh_hit* __restrict__ hits,
                                                        uint32_t* __restrict__ deep,     // [gridDim.x][kRayStackDeep][64]
                                                        uint32_t lds_depth,              // <= kRayStackLds
                                                        uint32_t deep_cap,               // <= kRayStackDeep (lbvh_debug_ray_stack_limit lowers it)
                                                        uint32_t* __restrict__ fault)    // mapped host word: a stack that ran out says so here
{
    __shared__ uint32_t s_stack[kRayStackLds][LBVH_WAVE];
    uint32_t* my_deep = deep + (size_t)blockIdx.x * (kRayStackDeep * LBVH_WAVE) + threadIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t total = *n_alive;
    // at least 32 rays per wave: with fewer live rays than lanes on the chip, half-filled waves on every wave slot hide more
    // latency than full waves on half of them (4 bounces: 1.487 -> 1.443 ms; 64: 1.487, 16: 1.449, 96 / 128 / 192: 1.71 / 1.91 / 2.43)
    const uint32_t run = max((total + gridDim.x - 1) / gridDim.x, 32u);
    uint32_t next = blockIdx.x * run;                      // scalar: next unclaimed entry of this wave's run
    if (next >= total) return;
    const uint32_t end = min(next + run, total);

    bool active = false;
    size_t i = 0;
    ray_t ray = {};
    float best_t = LBVH_MAX_FLOAT, best_u = 0.0f, best_v = 0.0f;
    uint32_t best_tri = 0, sp = 0, node = 0;
    for (;;) {
        // refill idle lanes from the chunk
        const uint64_t idle = __ballot(!active);
        if (idle != 0 && next < end) {
            if (!active) {
                const uint32_t k = next + mbcnt64(idle);
                if (k < end) {
                    i = list[k];
                    const float4* st = reinterpret_cast<const float4*>(&states[i]);
                    const float4 o = st[0], d = st[1];
                    ray.ox = o.x; ray.oy = o.y; ray.oz = o.z;
                    ray.dx = d.x; ray.dy = d.y; ray.dz = d.z;
                    ray.ix = 1.0f / d.x; ray.iy = 1.0f / d.y; ray.iz = 1.0f / d.z;
                    best_t = LBVH_MAX_FLOAT; best_tri = 0; best_u = 0.0f; best_v = 0.0f;
                    sp = 0; node = 0;
                    active = true;
                }
            }
            next += (uint32_t)__popcll(idle);
        }
        if (!__any(active)) break;
        if (active) {
            const float4* nb = reinterpret_cast<const float4*>(&nodes[node]);
            const float4 lmin = nb[0], lmax = nb[1], rmin = nb[2], rmax = nb[3];
            const uint32_t lref = __float_as_uint(lmin.w), rref = __float_as_uint(lmax.w);
            float tl, tr;
            bool hit_l = ray_box(lmin, lmax, ray, tl) && !(tl > best_t);
            bool hit_r = ray_box(rmin, rmax, ray, tr) && !(tr > best_t);
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const bool h = side == 0 ? hit_l : hit_r;
                const uint32_t ref = side == 0 ? lref : rref;
                if (h && (ref & 0x80000000u)) {
                    float4 v0, v1, v2;
                    unpack_fast_triangle(reinterpret_cast<const float4*>(&nodes[ref & 0x7FFFFFFFu]), v0, v1, v2);   // a triangle line
                    float u = 0.0f, v = 0.0f;
                    const float dist = ray_fast_triangle(ray, v0, v1, v2, u, v);
                    const uint32_t tri = __float_as_uint(v0.w);
                    if (dist > t_min && hit_counts(dist, side == 0 ? tl : tr) && (dist < best_t || (dist == best_t && tri < best_tri))) { best_t = dist; best_tri = tri; best_u = u; best_v = v; }
                }
            }
            const bool go_l = hit_l && !(lref & 0x80000000u) && !(tl > best_t);
            const bool go_r = hit_r && !(rref & 0x80000000u) && !(tr > best_t);
            if (go_l && go_r) {
                const bool l_near = tl <= tr;
                node = l_near ? lref : rref;
                const uint32_t far = l_near ? rref : lref;
                if (sp < lds_depth) { s_stack[sp][lane] = far; sp++; }
                else if (sp < lds_depth + deep_cap) { my_deep[(sp - lds_depth) * LBVH_WAVE] = far; sp++; }
                else __hip_atomic_store(fault, LBVH_FAULT_RAY_STACK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // never on this library's trees (see kWideStackDeep)
            } else if (go_l) {
                node = lref;
            } else if (go_r) {
                node = rref;
            } else if (sp != 0) {
                sp--;
                node = sp < lds_depth ? s_stack[sp][lane] : my_deep[(sp - lds_depth) * LBVH_WAVE];
            } else {
                float4 out;
                out.x = best_t;
                out.y = __uint_as_float(best_tri);
                out.z = best_u;
                out.w = best_v;
                reinterpret_cast<float4*>(hits)[i] = out;
                active = false;
            }
        }
    }
}

// ---- the per-ray walk over four-wide nodes -------------------------------------------------------------------
// The per-ray kernel's time follows the number of steps its lanes take (profiles/r3: requests, bytes and L1 traffic per
// step could be halved without moving it), so the derived scene gets a second, shallower form for rays that do not
// come in packets: every binary node with its larger children opened once or twice = up to four child boxes in one
// 128-byte line.  Same leaves, same boxes, same triangle lines: which triangles a ray can meet does not change, and
// with ties going to the lower triangle index neither does the record it ends with.
struct alignas(128) lbvh_wide_node {
    float lo[3][4];          // [axis][slot]
    float hi[3][4];
    uint32_t ref[4];         // node index | LEAF + triangle line | kWideEmpty
    uint32_t pad[4];
};
static_assert(sizeof(lbvh_wide_node) == 128, "wide node must be one 128-byte line");
constexpr uint32_t kWideEmpty = 0xFFFFFFFFu;

struct wide_slot { float mn[3], mx[3]; uint32_t ref; };

__device__ __forceinline__ float half_area(const wide_slot& b)
{
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ void children_of(const lbvh_fast_node* __restrict__ nodes, uint32_t i, wide_slot& l, wide_slot& r)
{
    const float4* q = reinterpret_cast<const float4*>(&nodes[i]);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    l.mn[0] = q0.x; l.mn[1] = q0.y; l.mn[2] = q0.z; l.mx[0] = q1.x; l.mx[1] = q1.y; l.mx[2] = q1.z; l.ref = __float_as_uint(q0.w);
    r.mn[0] = q2.x; r.mn[1] = q2.y; r.mn[2] = q2.z; r.mx[0] = q3.x; r.mx[1] = q3.y; r.mx[2] = q3.z; r.ref = __float_as_uint(q1.w);
}

// one thread per binary node: its two children, then twice the internal child with the largest surface opened
__global__ __launch_bounds__(256) void collapse_wide_kernel(const lbvh_fast_node* __restrict__ nodes, uint32_t n_internal,
                                                            lbvh_wide_node* __restrict__ wide)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_internal) return;
    wide_slot s[4];
    children_of(nodes, i, s[0], s[1]);
    s[2].ref = kWideEmpty; s[3].ref = kWideEmpty;
#pragma unroll
    for (int k = 2; k < 4; k++)
#pragma unroll
        for (int a = 0; a < 3; a++) { s[k].mn[a] = 0.0f; s[k].mx[a] = 0.0f; }
#pragma unroll
    for (int round = 0; round < 2; round++) {
        int open = -1;
        float open_area = -1.0f;
#pragma unroll
        for (int k = 0; k < 2 + round; k++) {
            const float a = half_area(s[k]);
            if (!(s[k].ref & 0x80000000u) && (open < 0 || a > open_area)) { open = k; open_area = a; }
        }
        if (open < 0) break;
        uint32_t parent = 0;
#pragma unroll
        for (int k = 0; k < 2 + round; k++)
            if (k == open) parent = s[k].ref;
        wide_slot l, r;
        children_of(nodes, parent, l, r);
#pragma unroll
        for (int k = 0; k < 2 + round; k++)
            if (k == open) s[k] = l;
        s[2 + round] = r;
    }
    lbvh_wide_node out;
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
        for (int a = 0; a < 3; a++) { out.lo[a][k] = s[k].mn[a]; out.hi[a][k] = s[k].mx[a]; }
        out.ref[k] = s[k].ref;
        out.pad[k] = 0u;
    }
    float4* dst = reinterpret_cast<float4*>(&wide[i]);
    const float4* src = reinterpret_cast<const float4*>(&out);
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k] = src[k];
}

__device__ __forceinline__ uint32_t pick4(const uint4 v, uint32_t k) { return k == 0u ? v.x : (k == 1u ? v.y : (k == 2u ? v.z : v.w)); }

// RayBoxIntersection (Raytracing.compute:75-87) on one slot of a wide node: the same six products, minima and maxima
__device__ __forceinline__ bool wide_box(float lx, float ly, float lz, float hx, float hy, float hz, const ray_t& r, float& tmin_out)
{
    return ray_box(make_float4(lx, ly, lz, 0.0f), make_float4(hx, hy, hz, 0.0f), r, tmin_out);
}

// what a launch of the per-ray walk did (lbvh_ray_stats_target: the algorithmic bytes of cfg5's roofline): wave reduction,
// one atomic per counter per wave at the end of the kernel
__device__ __forceinline__ void add_ray_stats(lbvh_ray_stats* stats, uint32_t rays, uint32_t steps, uint32_t tris)
{
    uint32_t v[3] = {rays, steps, tris};
#pragma unroll
    for (int k = 0; k < 3; k++) v[k] = wave_total_from_inclusive(wave_inclusive_sum(v[k]));
    if (lane_id() == 0) {
        atomicAdd((unsigned long long*)&stats->rays, (unsigned long long)v[0]);
        atomicAdd((unsigned long long*)&stats->node_fetches, (unsigned long long)v[1]);
        atomicAdd((unsigned long long*)&stats->triangle_tests, (unsigned long long)v[2]);
    }
}

// Same frame as trace_rays_kernel (lane refill from the wave's run of live rays, LDS + device-memory stack).
template <bool STATS>
__global__ __launch_bounds__(64) void trace_rays_wide_kernel(const lbvh_path_state* __restrict__ states, const uint32_t* __restrict__ n_alive,
                                                             const uint32_t* __restrict__ list, float t_min,
                                                             const lbvh_wide_node* __restrict__ wide,
                                                             const lbvh_fast_node* __restrict__ lines, lbvh_hit* __restrict__ hits,
                                                             uint32_t* __restrict__ deep,     // [gridDim.x][kWideStackDeep][64]
                                                             uint32_t lds_depth,              // <= kWideStackLds
                                                             uint32_t deep_cap,               // <= kWideStackDeep
                                                             uint32_t* __restrict__ fault, lbvh_ray_stats* stats)
{
    __shared__ uint32_t s_stack[kWideStackLds][LBVH_WAVE];
    uint32_t* my_deep = deep + (size_t)blockIdx.x * (kWideStackDeep * LBVH_WAVE) + threadIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t total = *n_alive;
    const uint32_t run = max((total + gridDim.x - 1) / gridDim.x, 32u);
    uint32_t next = blockIdx.x * run;
    if (next >= total) return;
    const uint32_t end = min(next + run, total);
    uint32_t n_rays = 0, n_steps = 0, n_tris = 0;

    bool active = false;
    size_t i = 0;
    ray_t ray = {};
    float best_t = LBVH_MAX_FLOAT, best_u = 0.0f, best_v = 0.0f;
    uint32_t best_tri = 0, sp = 0, node = 0;
    auto push = [&](uint32_t ref) {
        if (sp < lds_depth) { s_stack[sp][lane] = ref; sp++; }
        else if (sp < lds_depth + deep_cap) { my_deep[(sp - lds_depth) * LBVH_WAVE] = ref; sp++; }
        // a dropped entry would be a silently wrong hit: report it.  Cannot happen on this library's trees (a radix
        // tree over unique 32-bit keys is <= 32 levels deep, 3 waiting siblings per level); lbvh_debug_ray_stack_limit provokes it
        else __hip_atomic_store(fault, LBVH_FAULT_RAY_STACK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    for (;;) {
        const uint64_t idle = __ballot(!active);
        if (idle != 0 && next < end) {
            if (!active) {
                const uint32_t k = next + mbcnt64(idle);
                if (k < end) {
                    i = list[k];
                    const float4* st = reinterpret_cast<const float4*>(&states[i]);
                    const float4 o = st[0], d = st[1];
                    ray.ox = o.x; ray.oy = o.y; ray.oz = o.z;
                    ray.dx = d.x; ray.dy = d.y; ray.dz = d.z;
                    ray.ix = 1.0f / d.x; ray.iy = 1.0f / d.y; ray.iz = 1.0f / d.z;
                    best_t = LBVH_MAX_FLOAT; best_tri = 0; best_u = 0.0f; best_v = 0.0f;
                    sp = 0; node = 0;
                    active = true;
                    if (STATS) n_rays++;
                }
            }
            next += (uint32_t)__popcll(idle);
        }
        if (!__any(active)) break;
        if (active) {
            if (STATS) n_steps++;
            const float4* w = reinterpret_cast<const float4*>(&wide[node]);
            const float4 lox = w[0], loy = w[1], loz = w[2], hix = w[3], hiy = w[4], hiz = w[5];
            const uint4 ref = reinterpret_cast<const uint4*>(w)[6];
            float t0, t1, t2, t3;
            const bool h0 = wide_box(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ray, t0) && !(t0 > best_t) && ref.x != kWideEmpty;
            const bool h1 = wide_box(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ray, t1) && !(t1 > best_t) && ref.y != kWideEmpty;
            const bool h2 = wide_box(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ray, t2) && !(t2 > best_t) && ref.z != kWideEmpty;
            const bool h3 = wide_box(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ray, t3) && !(t3 > best_t) && ref.w != kWideEmpty;
            // leaf slots first: a lane's leaves one after the other, every lane's k-th at the same time
            uint32_t leaves = (h0 && (ref.x >> 31) ? 1u : 0u) | (h1 && (ref.y >> 31) ? 2u : 0u) | (h2 && (ref.z >> 31) ? 4u : 0u) |
                              (h3 && (ref.w >> 31) ? 8u : 0u);
            while (leaves != 0u) {
                const uint32_t k = (uint32_t)__builtin_ctz(leaves);
                leaves &= leaves - 1u;
                if (STATS) n_tris++;
                float4 v0, v1, v2;
                unpack_fast_triangle(reinterpret_cast<const float4*>(&lines[pick4(ref, k) & 0x7FFFFFFFu]), v0, v1, v2);
                float u = 0.0f, v = 0.0f;
                const float dist = ray_fast_triangle(ray, v0, v1, v2, u, v);
                const uint32_t tri = __float_as_uint(v0.w);
                const float entry = k == 0u ? t0 : (k == 1u ? t1 : (k == 2u ? t2 : t3));
                // ties go to the lower triangle index, whatever order the leaves are met in (as in the packet walk)
                if (dist > t_min && hit_counts(dist, entry) && (dist < best_t || (dist == best_t && tri < best_tri))) { best_t = dist; best_tri = tri; best_u = u; best_v = v; }
            }
            // nodes to enter, ordered by entry distance: the order key is the distance's bit pattern (non-negative floats
            // order like integers) with the slot number in its two lowest bits
            constexpr uint32_t none = 0xFFFFFFFFu;
            uint32_t k0 = h0 && !(ref.x >> 31) && !(t0 > best_t) ? ((__float_as_uint(fmaxf(t0, 0.0f)) & ~3u) | 0u) : none;
            uint32_t k1 = h1 && !(ref.y >> 31) && !(t1 > best_t) ? ((__float_as_uint(fmaxf(t1, 0.0f)) & ~3u) | 1u) : none;
            uint32_t k2 = h2 && !(ref.z >> 31) && !(t2 > best_t) ? ((__float_as_uint(fmaxf(t2, 0.0f)) & ~3u) | 2u) : none;
            uint32_t k3 = h3 && !(ref.w >> 31) && !(t3 > best_t) ? ((__float_as_uint(fmaxf(t3, 0.0f)) & ~3u) | 3u) : none;
            {   // five compare-exchanges
                uint32_t a, b;
                a = min(k0, k1); b = max(k0, k1); k0 = a; k1 = b;
                a = min(k2, k3); b = max(k2, k3); k2 = a; k3 = b;
                a = min(k0, k2); b = max(k0, k2); k0 = a; k2 = b;
                a = min(k1, k3); b = max(k1, k3); k1 = a; k3 = b;
                a = min(k1, k2); b = max(k1, k2); k1 = a; k2 = b;
            }
            if (k0 != none) {
                if (k3 != none) push(pick4(ref, k3 & 3u));       // farthest first: the nearest waiting sibling is popped first
                if (k2 != none) push(pick4(ref, k2 & 3u));
                if (k1 != none) push(pick4(ref, k1 & 3u));
                node = pick4(ref, k0 & 3u);
            } else if (sp != 0) {
                sp--;
                node = sp < lds_depth ? s_stack[sp][lane] : my_deep[(sp - lds_depth) * LBVH_WAVE];
            } else {
                float4 out;
                out.x = best_t;
                out.y = __uint_as_float(best_tri);
                out.z = best_u;
                out.w = best_v;
                reinterpret_cast<float4*>(hits)[i] = out;
                active = false;
            }
        }
    }
    if (STATS) add_ray_stats(stats, n_rays, n_steps, n_tris);
}

// The same walk for launches whose time is the chain of their longest rays (the later bounces of a frame: a few hundred
// thousand live rays, most of the chip idle for most of the launch): the next node is chosen BEFORE the step's triangles are
// tested and requested together with the first triangle line, so the two fetches of a step are in flight at once.  84 VGPRs
// (5 waves per SIMD instead of 8): the price where every wave slot is needed, none where they are not.
template <bool STATS>
__global__ __launch_bounds__(64) void trace_rays_wide_chain_kernel(const lbvh_path_state* __restrict__ states, const uint32_t* __restrict__ n_alive,
                                                             const uint32_t* __restrict__ list, float t_min,
                                                             const lbvh_wide_node* __restrict__ wide,
                                                             const lbvh_fast_node* __restrict__ lines, lbvh_hit* __restrict__ hits,
                                                             uint32_t* __restrict__ deep,     // [gridDim.x][kWideStackDeep][64]
                                                             uint32_t lds_depth,              // <= kWideStackLds
                                                             uint32_t deep_cap,               // <= kWideStackDeep
                                                             uint32_t* __restrict__ fault, lbvh_ray_stats* stats)
{
    __shared__ uint32_t s_stack[kWideStackLds][LBVH_WAVE];
    uint32_t* my_deep = deep + (size_t)blockIdx.x * (kWideStackDeep * LBVH_WAVE) + threadIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t total = *n_alive;
    const uint32_t run = max((total + gridDim.x - 1) / gridDim.x, 32u);
    uint32_t next = blockIdx.x * run;
    if (next >= total) return;
    const uint32_t end = min(next + run, total);
    uint32_t n_rays = 0, n_steps = 0, n_tris = 0;

    bool active = false, have = false;      // have: the registers below hold this lane's node
    uint32_t i = 0;                          // (lbvh_trace_rays: count <= 2^32 - 1)
    ray_t ray = {};
    float best_t = LBVH_MAX_FLOAT, best_u = 0.0f, best_v = 0.0f;
    uint32_t best_tri = 0, sp = 0, node = 0;
    float4 lox = {}, loy = {}, loz = {}, hix = {}, hiy = {}, hiz = {};
    uint4 ref = {};
    auto push = [&](uint32_t r) {
        if (sp < lds_depth) { s_stack[sp][lane] = r; sp++; }
        else if (sp < lds_depth + deep_cap) { my_deep[(sp - lds_depth) * LBVH_WAVE] = r; sp++; }
        // a dropped entry would be a silently wrong hit: report it.  Cannot happen on this library's trees (a radix
        // tree over unique 32-bit keys is <= 32 levels deep, 3 waiting siblings per level); lbvh_debug_ray_stack_limit provokes it
        else __hip_atomic_store(fault, LBVH_FAULT_RAY_STACK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    // A step: box tests on the node fetched during the previous step, the next node chosen (against the best hit so far),
    // then the first leaf's triangle line AND the next node requested together, the triangle tests while both are on their
    // way.  The two fetches of a step overlap instead of following each other; what a leaf hit would have pruned from the
    // choice is met by the next step's box tests (its children then fail `entry <= best`).
    for (;;) {
        const uint64_t idle = __ballot(!active);
        if (idle != 0 && next < end) {
            if (!active) {
                const uint32_t k = next + mbcnt64(idle);
                if (k < end) {
                    i = list[k];
                    const float4* st = reinterpret_cast<const float4*>(&states[i]);
                    const float4 o = st[0], d = st[1];
                    ray.ox = o.x; ray.oy = o.y; ray.oz = o.z;
                    ray.dx = d.x; ray.dy = d.y; ray.dz = d.z;
                    ray.ix = 1.0f / d.x; ray.iy = 1.0f / d.y; ray.iz = 1.0f / d.z;
                    best_t = LBVH_MAX_FLOAT; best_tri = 0; best_u = 0.0f; best_v = 0.0f;
                    sp = 0; node = 0;
                    active = true; have = false;
                    if (STATS) n_rays++;
                }
            }
            next += (uint32_t)__popcll(idle);
        }
        if (!__any(active)) break;
        uint32_t leaves = 0u;
        uint4 leaf_ref = {};
        float4 leaf_entry = {};
        bool fetch = active, done = false;
        if (active && have) {
            float t0, t1, t2, t3;
            const bool h0 = wide_box(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, ray, t0) && !(t0 > best_t) && ref.x != kWideEmpty;
            const bool h1 = wide_box(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, ray, t1) && !(t1 > best_t) && ref.y != kWideEmpty;
            const bool h2 = wide_box(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, ray, t2) && !(t2 > best_t) && ref.z != kWideEmpty;
            const bool h3 = wide_box(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, ray, t3) && !(t3 > best_t) && ref.w != kWideEmpty;
            leaves = (h0 && (ref.x >> 31) ? 1u : 0u) | (h1 && (ref.y >> 31) ? 2u : 0u) | (h2 && (ref.z >> 31) ? 4u : 0u) |
                     (h3 && (ref.w >> 31) ? 8u : 0u);
            leaf_ref = ref;
            leaf_entry = make_float4(t0, t1, t2, t3);
            // nodes to enter, ordered by entry distance: the order key is the distance's bit pattern (non-negative floats
            // order like integers) with the slot number in its two lowest bits
            constexpr uint32_t none = 0xFFFFFFFFu;
            uint32_t k0 = h0 && !(ref.x >> 31) ? ((__float_as_uint(fmaxf(t0, 0.0f)) & ~3u) | 0u) : none;
            uint32_t k1 = h1 && !(ref.y >> 31) ? ((__float_as_uint(fmaxf(t1, 0.0f)) & ~3u) | 1u) : none;
            uint32_t k2 = h2 && !(ref.z >> 31) ? ((__float_as_uint(fmaxf(t2, 0.0f)) & ~3u) | 2u) : none;
            uint32_t k3 = h3 && !(ref.w >> 31) ? ((__float_as_uint(fmaxf(t3, 0.0f)) & ~3u) | 3u) : none;
            {   // five compare-exchanges
                uint32_t a, b;
                a = min(k0, k1); b = max(k0, k1); k0 = a; k1 = b;
                a = min(k2, k3); b = max(k2, k3); k2 = a; k3 = b;
                a = min(k0, k2); b = max(k0, k2); k0 = a; k2 = b;
                a = min(k1, k3); b = max(k1, k3); k1 = a; k3 = b;
                a = min(k1, k2); b = max(k1, k2); k1 = a; k2 = b;
            }
            if (k0 != none) {
                if (k3 != none) push(pick4(ref, k3 & 3u));       // farthest first: the nearest waiting sibling is popped first
                if (k2 != none) push(pick4(ref, k2 & 3u));
                if (k1 != none) push(pick4(ref, k1 & 3u));
                node = pick4(ref, k0 & 3u);
            } else if (sp != 0) {
                sp--;
                node = sp < lds_depth ? s_stack[sp][lane] : my_deep[(sp - lds_depth) * LBVH_WAVE];
            } else {
                fetch = false;
                done = true;              // after this step's leaves
            }
        }
        // the first leaf's line, then the next node: requested back to back
        // (a triangle line: {v0, index} | e2.x in dword 7 | {e1, e2.y} | e2.z in dword 15)
        float4 q0 = {}, q1 = {}, q2 = {}, q3 = {};
        if (leaves != 0u) {
            const float4* line = reinterpret_cast<const float4*>(&lines[pick4(leaf_ref, (uint32_t)__builtin_ctz(leaves)) & 0x7FFFFFFFu]);
            q0 = line[0]; q1 = line[1]; q2 = line[2]; q3 = line[3];
        }
        if (fetch) {
            if (STATS) n_steps++;
            const float4* w = reinterpret_cast<const float4*>(&wide[node]);
            lox = w[0]; loy = w[1]; loz = w[2]; hix = w[3]; hiy = w[4]; hiz = w[5];
            ref = reinterpret_cast<const uint4*>(w)[6];
            have = true;
        }
        while (leaves != 0u) {
            const uint32_t slot = (uint32_t)__builtin_ctz(leaves);
            const float entry = slot == 0u ? leaf_entry.x : (slot == 1u ? leaf_entry.y : (slot == 2u ? leaf_entry.z : leaf_entry.w));
            leaves &= leaves - 1u;
            if (STATS) n_tris++;
            float u = 0.0f, v = 0.0f;
            const float dist = ray_triangle_edges(ray, q0, q2.x, q2.y, q2.z, q1.w, q2.w, q3.w, u, v);
            const uint32_t tri = __float_as_uint(q0.w);
            // ties go to the lower triangle index, whatever order the leaves are met in (as in the packet walk)
            if (dist > t_min && hit_counts(dist, entry) && (dist < best_t || (dist == best_t && tri < best_tri))) { best_t = dist; best_tri = tri; best_u = u; best_v = v; }
            if (leaves != 0u) {
                const float4* line = reinterpret_cast<const float4*>(&lines[pick4(leaf_ref, (uint32_t)__builtin_ctz(leaves)) & 0x7FFFFFFFu]);
                q0 = line[0]; q1 = line[1]; q2 = line[2]; q3 = line[3];
            }
        }
        if (done) {
            float4 out;
            out.x = best_t;
            out.y = __uint_as_float(best_tri);
            out.z = best_u;
            out.w = best_v;
            reinterpret_cast<float4*>(hits)[i] = out;
            active = false;
        }
    }
    if (STATS) add_ray_stats(stats, n_rays, n_steps, n_tris);
}

// ---- bounce ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v)
{
    const uint32_t state = v * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

__device__ __forceinline__ float path_rnd(uint32_t seed, uint32_t index, uint32_t bounce, uint32_t draw)
{
    const uint32_t h = pcg_hash(pcg_hash(pcg_hash(seed + 0x9E3779B9u * index) + bounce) + draw);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// one path's bounce; returns true when the path goes on
// MARK (lbvh_path_bounce, which owns the hit records between its calls): a path that ends leaves a DEAD record in
// hits[i] (t = MAX_FLOAT, triangle 0xFFFFFFFF), and a later bounce that finds it skips the path without touching its
// 64-byte state — after the first bounce most of a frame's paths are dead.
// FIRST (lbvh_path_first_bounce): the path state is not read but MADE here from the camera, as path_begin_kernel would
// have written it: the 132 MB of initial states of a 1080p frame are neither stored by one kernel nor loaded by the next.
template <bool MARK, bool FIRST>
__device__ __forceinline__ bool scatter_path(const lbvh_triangle* __restrict__ triangles, const lbvh_hit* hits, size_t i,
                                             uint32_t bounce, uint32_t seed, float albedo, lbvh_path_state* __restrict__ states,
                                             const lbvh_camera& cam);

// LIST: also append the indices of the paths that go on, so that the next segment is traced without a separate pass over
// all path states.  A workgroup handles kScatterItems x 256 paths and reserves its run of the list with ONE global atomic:
// with one path per thread the 8 100 workgroups of a 1080p frame queued on that one counter for most of the kernel's
// 67 - 91 us (the bounces' own work shrinks with the live paths, the kernel's time did not).
constexpr int kScatterItems = 8;
constexpr int kScatterItemsList = 2;       // ... when the paths come from a live list (see path_scatter_kernel)
// DOMAIN (round 5): the paths to look at are the `*n_domain` entries of `domain` — the list of live paths the previous bounce
// left behind — instead of all `count` pixels (domain == nullptr): from the second bounce on a scatter then costs what its live
// paths cost, not a pass over 2 M hit records of which most are dead (1080p: 35 / 33 / 33 us for bounces 2, 3 and the frame's last
// scatter -> 27 / 18 / 14 us).
// ITEMS paths per thread, one after the other (each item's ballot closes its chain of dependent loads): 8 for a pass over every
// pixel — few workgroups, one list reservation each —, 2 when the domain is a list of live paths: there the kernel is the latency
// of a thread's items in sequence (8 items: 32 us however few paths are live), not the reservation
template <bool LIST, bool FIRST = false, int ITEMS = kScatterItems>
__global__ __launch_bounds__(256) void path_scatter_kernel(const lbvh_triangle* __restrict__ triangles,
                                                           const lbvh_hit* hits, size_t count, uint32_t bounce,
                                                           uint32_t seed, float albedo, lbvh_path_state* __restrict__ states,
                                                           uint32_t* __restrict__ n_alive, uint32_t* __restrict__ list, lbvh_camera cam,
                                                           const uint32_t* __restrict__ domain, const uint32_t* __restrict__ n_domain)
{
    __shared__ uint32_t s_n, s_base;
    const size_t i0 = (size_t)blockIdx.x * (256 * ITEMS) + threadIdx.x;
    const size_t total = domain ? (size_t)*n_domain : count;
    if ((size_t)blockIdx.x * (256 * ITEMS) >= total) return;          // (uniform) nothing of the domain falls to this workgroup
    if (LIST) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    }
    uint32_t ofs[ITEMS];                      // position inside this wave's run, or ~0u: the path ended
    uint32_t idx[ITEMS];                      // the path (pixel) index of item k
    uint32_t wave_n = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const size_t at = i0 + (size_t)k * 256;
        const size_t i = at < total ? (domain ? (size_t)domain[at] : at) : 0;
        idx[k] = (uint32_t)i;
        const bool on = at < total && scatter_path<LIST, FIRST>(triangles, hits, i, bounce, seed, albedo, states, cam);
        if (LIST) {
            const uint64_t m = __ballot(on);
            ofs[k] = on ? wave_n + mbcnt64(m) : 0xFFFFFFFFu;
            wave_n += (uint32_t)__popcll(m);
        }
    }
    if (LIST) {
        uint32_t wave_base = 0;
        if (lane_id() == 0 && wave_n) wave_base = atomicAdd(&s_n, wave_n);
        wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_base);
        __syncthreads();
        if (threadIdx.x == 0 && s_n) s_base = __hip_atomic_fetch_add(n_alive, s_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; k++)
            if (ofs[k] != 0xFFFFFFFFu) list[s_base + wave_base + ofs[k]] = idx[k];
    }
}

template <bool MARK, bool FIRST>
__device__ __forceinline__ bool scatter_path(const lbvh_triangle* __restrict__ triangles, const lbvh_hit* hits, size_t i,
                                             uint32_t bounce, uint32_t seed, float albedo, lbvh_path_state* __restrict__ states,
                                             const lbvh_camera& cam)
{
    const float4 h = reinterpret_cast<const float4*>(hits)[i];
    const bool missed = !(h.x < LBVH_MAX_FLOAT);
    // ended in an earlier bounce: only lbvh_path_bounce writes DEAD records, so at bounce 0 a {MAX_FLOAT, 0xFFFFFFFF} record is
    // the caller's own fill value (an untraced pixel) and is an ordinary miss, as in lbvh_path_scatter
    if (MARK && bounce > 0u && missed && __float_as_uint(h.y) == 0xFFFFFFFFu) return false;
    float4* st = reinterpret_cast<float4*>(&states[i]);
    float4 o, d, thr, rad;
    if (FIRST) {                     // path_begin_kernel's state for pixel i, not read but made
        const uint32_t py = (uint32_t)(i / (size_t)cam.screen_width), px = (uint32_t)(i - (size_t)py * cam.screen_width);
        const ray_t r = make_ray(cam, px, py);
        o = make_float4(r.ox, r.oy, r.oz, __uint_as_float(1u));
        d = make_float4(r.dx, r.dy, r.dz, 0.0f);
        thr = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
        o = st[0];
        if (__float_as_uint(o.w) == 0u) return false;
        d = st[1]; thr = st[2]; rad = st[3];
    }
    if (missed) {
        const float sk = 0.5f * (d.y + 1.0f);
        rad.x = rad.x + thr.x * ((1.0f - sk) * 1.0f + sk * 0.5f);
        rad.y = rad.y + thr.y * ((1.0f - sk) * 1.0f + sk * 0.7f);
        rad.z = rad.z + thr.z * ((1.0f - sk) * 1.0f + sk * 1.0f);
        o.w = __uint_as_float(0u);
        st[0] = o;
        st[3] = rad;
        if (FIRST) { st[1] = d; st[2] = thr; }
        if (MARK) reinterpret_cast<float4*>(const_cast<lbvh_hit*>(hits))[i] = make_float4(LBVH_MAX_FLOAT, __uint_as_float(0xFFFFFFFFu), 0.0f, 0.0f);
        return false;
    }
    if (bounce == 0) rad.w = 1.0f;
    const float4* tp = reinterpret_cast<const float4*>(&triangles[__float_as_uint(h.y)]);
    const float4 a = tp[0], b = tp[1], c = tp[2];
    const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
    const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
    float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
    const float nl = sqrtf(dot3(nx, ny, nz, nx, ny, nz));
    if (nl > 0.0f) { nx = nx / nl; ny = ny / nl; nz = nz / nl; } else { nx = 0.0f; ny = 1.0f; nz = 0.0f; }
    if (dot3(nx, ny, nz, d.x, d.y, d.z) > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
    o.x = o.x + d.x * h.x; o.y = o.y + d.y * h.x; o.z = o.z + d.z * h.x;
    thr.x = thr.x * albedo; thr.y = thr.y * albedo; thr.z = thr.z * albedo;
    // uniform point on the unit sphere, Marsaglia 1972, at most 8 tries
    float px = 0.0f, py = 0.0f, pz = 1.0f;
    for (uint32_t k = 0; k < 16; k += 2) {
        const float x1 = 2.0f * path_rnd(seed, (uint32_t)i, bounce, k) - 1.0f;
        const float x2 = 2.0f * path_rnd(seed, (uint32_t)i, bounce, k + 1) - 1.0f;
        const float ss = x1 * x1 + x2 * x2;
        if (ss < 1.0f) {
            const float r = sqrtf(1.0f - ss);
            px = 2.0f * x1 * r; py = 2.0f * x2 * r; pz = 1.0f - 2.0f * ss;
            break;
        }
    }
    float vx = nx + px, vy = ny + py, vz = nz + pz;
    const float dl = sqrtf(dot3(vx, vy, vz, vx, vy, vz));
    if (dl > 1e-6f) { vx = vx / dl; vy = vy / dl; vz = vz / dl; } else { vx = nx; vy = ny; vz = nz; }
    d.x = vx; d.y = vy; d.z = vz;
    st[0] = o; st[1] = d; st[2] = thr; st[3] = rad;
    return true;
}

__global__ __launch_bounds__(256) void path_resolve_kernel(const lbvh_path_state* __restrict__ states, size_t count,
                                                           uint16_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4 rad = reinterpret_cast<const float4*>(&states[i])[3];
    ushort4 o;
    o.x = __half_as_ushort(__float2half_rn(rad.x));
    o.y = __half_as_ushort(__float2half_rn(rad.y));
    o.z = __half_as_ushort(__float2half_rn(rad.z));
    o.w = __half_as_ushort(__float2half_rn(rad.w));
    reinterpret_cast<ushort4*>(out)[i] = o;
}

}  // namespace

static_assert(sizeof(lbvh_path_state) == 64, "path state must be 64 bytes");

// the walk over the live rays of `list`: four-wide nodes (made on first use after a rebuild) — few_rays: with the kernel that
// keeps two fetches of a step in flight (the later bounces of a frame) —, or the binary nodes the packet walk uses
// (lbvh_debug_ray_walker(ctx, 0): the cross-check of the tests; 2: the few-rays kernel for every launch)
static lbvh_status launch_ray_walk(lbvh_context* ctx, const lbvh_path_state* d_states, const uint32_t* n_alive, const uint32_t* list,
                                   float t_min, lbvh_hit* d_hits, size_t count, bool few_rays = false)
{
    const uint32_t ray_waves = ray_waves_of(count);
    if (ctx->ray_walker != 0u) {
        const uint32_t n_internal = ctx->fast_src.n - 1;
        if (!ctx->wide_valid) {
            const int rc = lbvh_reserve(ctx, &ctx->wide_nodes, &ctx->wide_nodes_bytes, (size_t)n_internal * sizeof(lbvh_wide_node));
            if (rc != LBVH_OK) return (lbvh_status)rc;
            LBVH_LAUNCH(ctx, collapse_wide_kernel, dim3((n_internal + 255) / 256), dim3(256), ctx->fast_nodes, n_internal,
                        (lbvh_wide_node*)ctx->wide_nodes);
            ctx->wide_valid = true;
        }
        const uint32_t lds = std::min<uint32_t>(ctx->ray_stack_lds, kWideStackLds), deep_cap = std::min<uint32_t>(ctx->ray_stack_deep, kWideStackDeep);
        lbvh_ray_stats* st = ctx->ray_stats;
        const lbvh_wide_node* wn = (const lbvh_wide_node*)ctx->wide_nodes;
        if (few_rays || ctx->ray_walker == 2u) {
            if (st) LBVH_LAUNCH(ctx, trace_rays_wide_chain_kernel<true>, dim3(ray_waves), dim3(LBVH_WAVE), d_states, n_alive, list, t_min, wn, ctx->fast_nodes,
                                d_hits, deep_stacks(ctx, count), lds, deep_cap, ctx->fault_dev, st);
            else LBVH_LAUNCH(ctx, trace_rays_wide_chain_kernel<false>, dim3(ray_waves), dim3(LBVH_WAVE), d_states, n_alive, list, t_min, wn, ctx->fast_nodes,
                             d_hits, deep_stacks(ctx, count), lds, deep_cap, ctx->fault_dev, st);
        } else {
            if (st) LBVH_LAUNCH(ctx, trace_rays_wide_kernel<true>, dim3(ray_waves), dim3(LBVH_WAVE), d_states, n_alive, list, t_min, wn, ctx->fast_nodes,
                                d_hits, deep_stacks(ctx, count), lds, deep_cap, ctx->fault_dev, st);
            else LBVH_LAUNCH(ctx, trace_rays_wide_kernel<false>, dim3(ray_waves), dim3(LBVH_WAVE), d_states, n_alive, list, t_min, wn, ctx->fast_nodes,
                             d_hits, deep_stacks(ctx, count), lds, deep_cap, ctx->fault_dev, st);
        }
    } else {
        LBVH_LAUNCH(ctx, trace_rays_kernel, dim3(ray_waves), dim3(LBVH_WAVE), d_states, n_alive, list, t_min, ctx->fast_nodes,
                    ctx->fast_tris, d_hits, deep_stacks(ctx, count), ctx->ray_stack_lds, std::min<uint32_t>(ctx->ray_stack_deep, kRayStackDeep), ctx->fault_dev);
    }
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}


extern "C" {

lbvh_status lbvh_animate(lbvh_context* ctx, const lbvh_triangle* d_rest, uint32_t n, const uint32_t* d_body,
                         const float* d_centres, float cos_angle, float sin_angle, lbvh_triangle* d_out)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (n == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_rest != nullptr && d_body != nullptr && d_centres != nullptr && d_out != nullptr);
    LBVH_REQUIRE(ctx, d_rest != d_out);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_rest & 15) == 0 && ((uintptr_t)d_out & 15) == 0 && ((uintptr_t)d_centres & 15) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    lbvh_note_write(ctx, d_out, (size_t)n * sizeof(lbvh_triangle));
    LBVH_LAUNCH(ctx, animate_kernel, dim3((unsigned)(((size_t)n * 8 + 255) / 256)), dim3(256), d_rest, n, d_body, (const float4*)d_centres, cos_angle,
                sin_angle, d_out);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

lbvh_status lbvh_path_begin(lbvh_context* ctx, const lbvh_camera* h_camera, lbvh_path_state* d_states)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, h_camera != nullptr && d_states != nullptr && ((uintptr_t)d_states & 15) == 0);
    const lbvh_camera cam = *h_camera;
    LBVH_REQUIRE(ctx, cam.screen_width > 0 && cam.screen_height > 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->ray_list.valid = false;                      // new states: no bounce has listed their live paths yet
    LBVH_LAUNCH(ctx, path_begin_kernel, dim3((cam.screen_width + 15) / 16, (cam.screen_height + 15) / 16), dim3(256), cam,
                d_states);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

lbvh_status lbvh_trace_rays(lbvh_context* ctx, const lbvh_path_state* d_states, size_t count, float t_min,
                            const lbvh_scene* h_scene, lbvh_hit* d_hits)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (count == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_states != nullptr && h_scene != nullptr && d_hits != nullptr);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_states & 15) == 0 && ((uintptr_t)d_hits & 15) == 0);
    LBVH_REQUIRE(ctx, count <= 0xFFFFFFFFull);
    {
        const int frc = lbvh_require_fast(ctx, *h_scene, "lbvh_trace_rays");
        if (frc != LBVH_OK) return frc;
    }
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // scratch: [live-ray count (256 B) | indices of the live rays]
    int rc = lbvh_reserve(ctx, &ctx->ray_scratch, &ctx->ray_scratch_bytes, ray_scratch_bytes_for(count));
    if (rc != LBVH_OK) return rc;
    ctx->ray_list.valid = false;                      // this call takes list 0 for its own rays
    uint32_t* n_alive = ray_counter(ctx, 0);
    uint32_t* list = ray_list(ctx, count, 0);
    LBVH_HIP_TRY(ctx, hipMemsetAsync(n_alive, 0, 4, ctx->cur_stream));
    LBVH_LAUNCH(ctx, alive_rays_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), d_states, count, n_alive, list, d_hits);
    return launch_ray_walk(ctx, d_states, n_alive, list, t_min, d_hits, count);
}

lbvh_status lbvh_debug_ray_walker(lbvh_context* ctx, uint32_t walker)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, walker <= 2u);
    ctx->ray_walker = walker;
    return LBVH_OK;
}

lbvh_status lbvh_debug_ray_stack_split(lbvh_context* ctx, uint32_t lds_entries)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, lds_entries >= 1 && lds_entries <= (uint32_t)kRayStackLds);
    ctx->ray_stack_lds = lds_entries;
    return LBVH_OK;
}

lbvh_status lbvh_ray_stats_target(lbvh_context* ctx, lbvh_ray_stats* d_stats)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, ((uintptr_t)d_stats & 7) == 0);
    ctx->ray_stats = d_stats;
    return LBVH_OK;
}

lbvh_status lbvh_debug_ray_stack_limit(lbvh_context* ctx, uint32_t deep_entries)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    ctx->ray_stack_deep = deep_entries == 0u ? 0xFFFFFFFFu : deep_entries;       // the kernels clamp it to their slab's size
    return LBVH_OK;
}

lbvh_status lbvh_path_scatter(lbvh_context* ctx, const lbvh_scene* h_scene, const lbvh_hit* d_hits, size_t count,
                              uint32_t bounce, uint32_t seed, float albedo, lbvh_path_state* d_states)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (count == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, h_scene != nullptr && h_scene->triangles != nullptr && d_hits != nullptr && d_states != nullptr);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the live paths of the bounce before, if this call continues that very frame (same buffers, next bounce): only they can
    // have anything to scatter
    const bool from_list = bounce >= 2u && ctx->ray_list.valid && ctx->ray_list.states == (const void*)d_states && ctx->ray_list.hits == (const void*)d_hits &&
                           ctx->ray_list.count == count && ctx->ray_list.bounce + 1u == bounce && ctx->ray_scratch != nullptr;
    ctx->ray_list.valid = false;                      // the states change without a new list being made
    if (from_list)
        LBVH_LAUNCH(ctx, (path_scatter_kernel<false, false, kScatterItemsList>), dim3((unsigned)((count + 256 * kScatterItemsList - 1) / (256 * kScatterItemsList))),
                    dim3(256), h_scene->triangles, d_hits, count, bounce, seed, albedo, d_states, nullptr, nullptr, lbvh_camera{},
                    ray_list(ctx, count, ctx->ray_list.turn), ray_counter(ctx, ctx->ray_list.turn));
    else
        LBVH_LAUNCH(ctx, path_scatter_kernel<false>, dim3((unsigned)((count + 256 * kScatterItems - 1) / (256 * kScatterItems))), dim3(256), h_scene->triangles,
                    d_hits, count, bounce, seed, albedo, d_states, nullptr, nullptr, lbvh_camera{}, nullptr, nullptr);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

}  // extern "C"

static lbvh_status path_bounce_impl(lbvh_context* ctx, const lbvh_scene* h_scene, lbvh_path_state* d_states, lbvh_hit* d_hits,
                                    size_t count, uint32_t bounce, uint32_t seed, float albedo, float t_min,
                                    const lbvh_camera* h_first_camera)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (count == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, h_scene != nullptr && h_scene->triangles != nullptr && d_hits != nullptr && d_states != nullptr);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_states & 15) == 0 && ((uintptr_t)d_hits & 15) == 0 && count <= 0xFFFFFFFFull);
    {
        const int frc = lbvh_require_fast(ctx, *h_scene, "lbvh_path_bounce");
        if (frc != LBVH_OK) return frc;
    }
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* before = ctx->ray_scratch;
    int rc = lbvh_reserve(ctx, &ctx->ray_scratch, &ctx->ray_scratch_bytes, ray_scratch_bytes_for(count));
    if (rc != LBVH_OK) return rc;
    if (ctx->ray_scratch != before) ctx->ray_list.valid = false;
    // the paths to look at: the list the bounce before left behind, if this call continues that very frame (same buffers, next
    // bounce, nothing written into them since: lbvh_note_write); otherwise every pixel.  The new list goes into the other buffer.
    // (from the second bounce on: at bounce 1 two fifths of a frame's paths are still alive and the two forms cost the same, 53 - 56 us
    // at 1080p; bounces 2, 3 and the last scatter 35 / 33 / 33 -> 27 - 31 / 18 - 20 / 14 - 16 us; 4 paths per thread instead of 2: 33 / 21 / 20)
    const bool from_list = !h_first_camera && bounce >= 2u && ctx->ray_list.valid && ctx->ray_list.states == (const void*)d_states &&
                           ctx->ray_list.hits == (const void*)d_hits && ctx->ray_list.count == count && ctx->ray_list.bounce + 1u == bounce;
    const uint32_t prev = ctx->ray_list.turn, turn = from_list ? prev ^ 1u : 0u;
    uint32_t* n_alive = ray_counter(ctx, turn);
    uint32_t* list = ray_list(ctx, count, turn);
    LBVH_HIP_TRY(ctx, hipMemsetAsync(n_alive, 0, 4, ctx->cur_stream));
    const dim3 scatter_grid((unsigned)((count + 256 * kScatterItems - 1) / (256 * kScatterItems)));
    if (h_first_camera)
        LBVH_LAUNCH(ctx, (path_scatter_kernel<true, true>), scatter_grid, dim3(256), h_scene->triangles, d_hits, count, 0u, seed, albedo, d_states,
                    n_alive, list, *h_first_camera, nullptr, nullptr);
    else if (from_list)
        LBVH_LAUNCH(ctx, (path_scatter_kernel<true, false, kScatterItemsList>), dim3((unsigned)((count + 256 * kScatterItemsList - 1) / (256 * kScatterItemsList))),
                    dim3(256), h_scene->triangles, d_hits, count, bounce, seed, albedo, d_states, n_alive, list, lbvh_camera{}, ray_list(ctx, count, prev),
                    ray_counter(ctx, prev));
    else
        LBVH_LAUNCH(ctx, (path_scatter_kernel<true, false>), scatter_grid, dim3(256), h_scene->triangles, d_hits, count, bounce, seed, albedo,
                    d_states, n_alive, list, lbvh_camera{}, nullptr, nullptr);
    ctx->ray_list.valid = true;
    ctx->ray_list.states = d_states; ctx->ray_list.hits = d_hits; ctx->ray_list.count = count;
    ctx->ray_list.bounce = h_first_camera ? 0u : bounce;
    ctx->ray_list.turn = turn;
    return launch_ray_walk(ctx, d_states, n_alive, list, t_min, d_hits, count, bounce >= 1u);
}

extern "C" {

lbvh_status lbvh_path_bounce(lbvh_context* ctx, const lbvh_scene* h_scene, lbvh_path_state* d_states, lbvh_hit* d_hits,
                             size_t count, uint32_t bounce, uint32_t seed, float albedo, float t_min)
{
    return path_bounce_impl(ctx, h_scene, d_states, d_hits, count, bounce, seed, albedo, t_min, nullptr);
}

lbvh_status lbvh_path_first_bounce(lbvh_context* ctx, const lbvh_camera* h_camera, const lbvh_scene* h_scene, lbvh_path_state* d_states,
                                   lbvh_hit* d_hits, uint32_t seed, float albedo, float t_min)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    LBVH_REQUIRE(ctx, h_camera != nullptr && h_camera->screen_width > 0 && h_camera->screen_height > 0);
    return path_bounce_impl(ctx, h_scene, d_states, d_hits, (size_t)h_camera->screen_width * (size_t)h_camera->screen_height, 0u, seed, albedo,
                            t_min, h_camera);
}

lbvh_status lbvh_path_resolve(lbvh_context* ctx, const lbvh_path_state* d_states, size_t count, uint16_t* d_rgba16f)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (count == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_states != nullptr && d_rgba16f != nullptr && ((uintptr_t)d_rgba16f & 7) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    LBVH_LAUNCH(ctx, path_resolve_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), d_states, count, d_rgba16f);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

}  // extern "C"
