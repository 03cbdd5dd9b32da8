// Is a short reciprocal bit-identical to the IEEE-exact 1.0f / x for EVERY float?  All 2^32 bit patterns are compared
// (exhaustive, seconds on the GPU): the triangle test's `inv_det = 1.0 / det` (Raytracing.compute:50) must round like
// the oracle's C division, and the compiler's exact sequence is 11 instructions (v_div_scale x2, v_rcp, 4 fma, v_mul,
// v_div_fmas, v_div_fixup).  Candidates: v_rcp_f32 + one / two Newton steps in fma.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ float cand1(float x) { float r = __builtin_amdgcn_rcpf(x); const float e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float cand2(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(e, r, r);
}

__global__ void check(unsigned long long* bad, uint32_t* first_bad)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;      // 2^24 threads x 256 patterns
    unsigned long long b1 = 0, b2 = 0, n = 0;
    for (uint32_t k = 0; k < 256u; k++) {
        const uint32_t bits = (k << 24) | tid;                        // exponent / sign in the top byte varies with k
        const float x = __uint_as_float(bits);
        const float ax = fabsf(x);
        if (!(ax >= 1e-20f && ax <= 1e20f)) continue;                 // the range the walker would use the short form in
        n++;
        const float exact = 1.0f / x;
        if (__float_as_uint(cand1(x)) != __float_as_uint(exact)) { b1++; atomicMin(&first_bad[0], bits); }
        if (__float_as_uint(cand2(x)) != __float_as_uint(exact)) { b2++; atomicMin(&first_bad[1], bits); }
    }
    atomicAdd(&bad[0], b1); atomicAdd(&bad[1], b2); atomicAdd(&bad[2], n);
}

int main()
{
    unsigned long long* bad; uint32_t* fb;
    hipMalloc(&bad, 24); hipMalloc(&fb, 8);
    hipMemset(bad, 0, 24); hipMemset(fb, 0xFF, 8);
    check<<<1 << 16, 256>>>(bad, fb);
    unsigned long long h[3]; uint32_t f[2];
    hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost); hipMemcpy(f, fb, 8, hipMemcpyDeviceToHost);
    printf("floats with 1e-20 <= |x| <= 1e20: %llu\n", h[2]);
    printf("v_rcp + 1 Newton step  != 1.0f / x on %llu of them (first bits 0x%08x)\n", h[0], f[0]);
    printf("v_rcp + 2 Newton steps != 1.0f / x on %llu of them (first bits 0x%08x)\n", h[1], f[1]);
    return 0;
}
