// Is a product taken by the matrix pipe the product v_mul_f32 gives?  v_mfma_f32_4x4x1_16b_f32 with C = 0: within every block
// of four lanes, lane n receives A[m] * B[n] for the four A values of its block (m = 0 .. 3) — an outer product: what the packet
// walk's slab tests are ((plane - origin) of four planes x the lane's inverse direction).  Compared bit for bit with the
// products of v_mul_f32 on random bit patterns (normal, tiny, huge, denormal inputs and results, zeros, infinities).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ uint32_t hash(uint32_t v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }

// class 0: any bit pattern; 1: normal floats of moderate size; 2: tiny x tiny (denormal / underflowing products); 3: huge x huge
__device__ float sample(uint32_t h, int cls)
{
    uint32_t bits = h;
    if (cls == 1) bits = (h & 0x807FFFFFu) | ((100u + (h >> 23) % 56u) << 23);
    if (cls == 2) bits = (h & 0x807FFFFFu) | ((40u + (h >> 23) % 30u) << 23);
    if (cls == 3) bits = (h & 0x807FFFFFu) | ((180u + (h >> 23) % 70u) << 23);
    return __uint_as_float(bits);
}

__global__ void check(uint32_t seed, int cls, unsigned long long* out)    // out: [0] products, [1] mismatches, [2] of them +-0 only, [3] denormal results, [4] NaN payloads
{
    const uint32_t lane = threadIdx.x & 63u, g = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n = 0, bad = 0, zero_sign = 0, den = 0, nan = 0;
    for (uint32_t it = 0; it < 256; it++) {
        const float a = sample(hash(seed + g * 977u + it * 2654435761u), cls);
        const float b = sample(hash(seed * 31u + g * 7919u + it * 40503u + 17u), cls);
        const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        const f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, zero, 0, 0, 0);
        float am[4];
#pragma unroll
        for (int m = 0; m < 4; m++) am[m] = __shfl(a, (lane & ~3u) + m);       // A[m] of my block
#pragma unroll
        for (int m = 0; m < 4; m++) {
            float p;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(am[m]), "v"(b));
            const uint32_t x = __float_as_uint(d[m]), y = __float_as_uint(p);
            n++;
            if (x != y) {
                if (p != p && d[m] != d[m]) { nan++; continue; }
                bad++;
                if ((x | y) == 0x80000000u) zero_sign++;
                if ((y & 0x7F800000u) == 0 && (y & 0x007FFFFFu) != 0) den++;
            }
        }
    }
    atomicAdd(&out[0], n); atomicAdd(&out[1], bad); atomicAdd(&out[2], zero_sign); atomicAdd(&out[3], den); atomicAdd(&out[4], nan);
}

int main()
{
    unsigned long long* d;
    CK(hipMalloc(&d, 5 * 8));
    const char* names[4] = {"any bit patterns", "normal floats, moderate exponents", "tiny x tiny (denormal products)", "huge x huge (overflow)"};
    for (int cls = 0; cls < 4; cls++) {
        CK(hipMemset(d, 0, 5 * 8));
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, 12345u + cls, cls, d);
        unsigned long long h[5];
        CK(hipMemcpy(h, d, 5 * 8, hipMemcpyDeviceToHost));
        printf("%-36s products %llu  mismatches %llu (of them: +0 / -0 only %llu, v_mul result denormal %llu)  NaN-vs-NaN payload differences %llu\n",
               names[cls], h[0], h[1], h[2], h[3], h[4]);
    }
    return 0;
}
