// lbvh_shade.hip — the shading tail of the reference's Raytracing kernel
// (Assets/_Shaders/Raytracing/Raytracing.compute:178-184), one thread per RaycastResult.
// Strict fp32 in the reference's operation order (the library is built with -ffp-contract=off), so the
// RGBA16F output is bit-identical to the CPU oracle.
#include "lbvh_common.h"

namespace {

__device__ __forceinline__ float4 texel(const uint8_t* __restrict__ tex, int w, int x, int y)
{
    const uchar4 c = reinterpret_cast<const uchar4*>(tex)[(size_t)y * w + x];
    return make_float4((float)c.x / 255.0f, (float)c.y / 255.0f, (float)c.z / 255.0f, (float)c.w / 255.0f);
}

// SampleLevel(linearClampSampler, uv, 0): bilinear on texel centres, clamp addressing
__device__ __forceinline__ float4 sample_bilinear_clamp(const uint8_t* __restrict__ tex, int w, int h, float u, float v)
{
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    // clamp in float first: u, v may be far outside [0, 1]
    const float xc0 = fminf(fmaxf(xf, 0.0f), (float)(w - 1)), xc1 = fminf(fmaxf(xf + 1.0f, 0.0f), (float)(w - 1));
    const float yc0 = fminf(fmaxf(yf, 0.0f), (float)(h - 1)), yc1 = fminf(fmaxf(yf + 1.0f, 0.0f), (float)(h - 1));
    const int x0 = (int)xc0, x1 = (int)xc1, y0 = (int)yc0, y1 = (int)yc1;
    const float4 c00 = texel(tex, w, x0, y0), c10 = texel(tex, w, x1, y0), c01 = texel(tex, w, x0, y1),
                 c11 = texel(tex, w, x1, y1);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    float4 r;
    r.x = (c00.x * gx + c10.x * fx) * gy + (c01.x * gx + c11.x * fx) * fy;
    r.y = (c00.y * gx + c10.y * fx) * gy + (c01.y * gx + c11.y * fx) * fy;
    r.z = (c00.z * gx + c10.z * fx) * gy + (c01.z * gx + c11.z * fx) * fy;
    r.w = (c00.w * gx + c10.w * fx) * gy + (c01.w * gx + c11.w * fx) * fy;
    return r;
}

__global__ __launch_bounds__(256) void shade_kernel(const lbvh_hit* __restrict__ hits, size_t count,
                                                    const lbvh_triangle* __restrict__ tris,
                                                    const uint8_t* __restrict__ tex, int tex_w, int tex_h,
                                                    uint16_t* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4 hr = reinterpret_cast<const float4*>(hits)[i];
    const float dist = hr.x, u = hr.z, v = hr.w;
    const uint32_t tri = __float_as_uint(hr.y);
    const float4* tp = reinterpret_cast<const float4*>(&tris[tri]);     // Raytracing.compute:178
    const float4 uv0 = tp[3], uv1 = tp[4];       // a_uv, b_uv | c_uv, pad
    const float4 an = tp[5], bn = tp[6], cn = tp[7];
    const float wgt = (1.0f - u) - v;                                   // (1 - uv.x - uv.y)
    const float tu = (wgt * uv0.x + u * uv0.z) + v * uv1.x;            // :179
    const float tv = (wgt * uv0.y + u * uv0.w) + v * uv1.y;
    const float nx = (wgt * an.x + u * bn.x) + v * cn.x;               // :180
    const float ny = (wgt * an.y + u * bn.y) + v * cn.y;
    const float nz = (wgt * an.z + u * bn.z) + v * cn.z;
    const float light_dir = 0.57735026f;                                // :181 — a scalar
    const float lambert = fmaxf(0.4f, (light_dir * nx + light_dir * ny) + light_dir * nz);
    const float4 c = sample_bilinear_clamp(tex, tex_w, tex_h, tu, tv);  // :183
    const float alpha = dist != LBVH_MAX_FLOAT ? 1.0f : 0.0f;           // :184
    const __half h0 = __float2half_rn(c.x * lambert), h1 = __float2half_rn(c.y * lambert),
                 h2 = __float2half_rn(c.z * lambert), h3 = __float2half_rn(alpha);
    ushort4 o;
    o.x = __half_as_ushort(h0); o.y = __half_as_ushort(h1); o.z = __half_as_ushort(h2); o.w = __half_as_ushort(h3);
    reinterpret_cast<ushort4*>(out)[i] = o;
}

}  // namespace

extern "C" lbvh_status lbvh_shade(lbvh_context* ctx, const lbvh_hit* d_hits, size_t count,
                                  const lbvh_triangle* d_triangles, const uint8_t* d_texture_rgba8, int32_t tex_w,
                                  int32_t tex_h, uint16_t* d_rgba16f)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (count == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_hits != nullptr && d_triangles != nullptr && d_texture_rgba8 != nullptr && d_rgba16f != nullptr);
    LBVH_REQUIRE(ctx, tex_w > 0 && tex_h > 0);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_hits & 15) == 0 && ((uintptr_t)d_triangles & 15) == 0 &&
                          ((uintptr_t)d_texture_rgba8 & 3) == 0 && ((uintptr_t)d_rgba16f & 7) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t blocks = (count + 255) / 256;
    LBVH_LAUNCH(ctx, shade_kernel, dim3((unsigned)blocks), dim3(256), d_hits, count, d_triangles, d_texture_rgba8, tex_w, tex_h,
                d_rgba16f);
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}

// ---- Hidden/ImageComposer (Assets/_Shaders/ImageComposer.shader:44-52) ------------------------------------------
namespace {

__global__ __launch_bounds__(256) void compose_kernel(const uint2* __restrict__ background, const uint2* __restrict__ object,
                                                      size_t count, uint2* out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint2 b = background[i], o = object[i];       // 4 halfs each
    const __half2 b01 = *reinterpret_cast<const __half2*>(&b.x), b23 = *reinterpret_cast<const __half2*>(&b.y);
    const __half2 o01 = *reinterpret_cast<const __half2*>(&o.x), o23 = *reinterpret_cast<const __half2*>(&o.y);
    const float bg[3] = {__low2float(b01), __high2float(b01), __low2float(b23)};
    const float ob[3] = {__low2float(o01), __high2float(o01), __low2float(o23)};
    const float a = __high2float(o23);
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; k++) r[k] = bg[k] + a * (ob[k] - bg[k]);                         // lerp, :49
    const __half h0 = __float2half_rn(r[0]), h1 = __float2half_rn(r[1]), h2 = __float2half_rn(r[2]),
                 h3 = __float2half_rn(1.0f);                                                // fixed4(ret, 1), :52
    uint2 w;
    w.x = (uint32_t)__half_as_ushort(h0) | ((uint32_t)__half_as_ushort(h1) << 16);
    w.y = (uint32_t)__half_as_ushort(h2) | ((uint32_t)__half_as_ushort(h3) << 16);
    out[i] = w;
}

}  // namespace

extern "C" lbvh_status lbvh_compose(lbvh_context* ctx, const uint16_t* d_background_rgba16f, const uint16_t* d_object_rgba16f,
                                    size_t count, uint16_t* d_out_rgba16f)
{
    if (!ctx) return LBVH_ERR_INVALID_ARG;
    if (count == 0) return LBVH_OK;
    LBVH_REQUIRE(ctx, d_background_rgba16f != nullptr && d_object_rgba16f != nullptr && d_out_rgba16f != nullptr);
    LBVH_REQUIRE(ctx, ((uintptr_t)d_background_rgba16f & 7) == 0 && ((uintptr_t)d_object_rgba16f & 7) == 0 &&
                          ((uintptr_t)d_out_rgba16f & 7) == 0);
    LBVH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t blocks = (count + 255) / 256;
    LBVH_LAUNCH(ctx, compose_kernel, dim3((unsigned)blocks), dim3(256), reinterpret_cast<const uint2*>(d_background_rgba16f),
                reinterpret_cast<const uint2*>(d_object_rgba16f), count, reinterpret_cast<uint2*>(d_out_rgba16f));
    LBVH_HIP_TRY(ctx, hipGetLastError());
    return LBVH_OK;
}
