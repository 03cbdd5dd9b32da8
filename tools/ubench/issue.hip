// Issue-throughput microbenchmark: how do VALU / v_readlane(SGPR write) / ballot+scalar-branch mixes scale
// with waves per SIMD on gfx950?  No memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a = threadIdx.x * 0.001f + seed, b = a * 1.5f, c = b + 2.0f, d = c * 0.7f;
    int acc = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 64 VALU, VGPR operands only
#pragma unroll
            for (int j = 0; j < 16; j++) { a = a * b + c; b = b * c + d; c = c * d + a; d = d * a + b; }
        } else if (MODE == 1) {   // 48 VALU + 16 readlane->SGPR feeding VALU as scalar operands
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), j));
                a = a * s + c; b = b * s + d; c = c * s + a;
            }
        } else if (MODE == 2) {   // 48 VALU + 8 ballots with scalar branches on them
#pragma unroll
            for (int j = 0; j < 8; j++) {
                a = a * b + c; b = b * c + d; c = c * d + a; d = d * a + b; a = a * b + d; b = b + c;
                const unsigned long long m = __ballot(a > b);
                if (m == 0x123456789ull) { acc += 1; a += 1.0f; }   // never true, but a real scalar branch
            }
        } else {   // MODE 3: 48 VALU + 8 exec-masked regions (s_and_saveexec / s_or exec / cbranch_execz)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                a = a * b + c; b = b * c + d; c = c * d + a; d = d * a + b; a = a * b + d; b = b + c;
                if (a > b + (float)j) { c = c * 1.0001f + d; }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + acc;
}

template <int MODE>
void run(float* out, const char* name)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int bpc : {1, 2, 4, 8}) {
        const int blocks = 256 * bpc;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s waves/SIMD=%d  %8.3f ms  cycles/iter/wave(at 2.4GHz)=%8.1f  iters/us/CU=%8.2f\n", name, bpc, ms,
               ms * 1e-3 * 2.4e9 / iters, (double)iters * bpc * 4 / (ms * 1e3));
    }
}

int main()
{
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>(out, "64 VALU (vgpr only)");
    run<1>(out, "48 VALU + 16 readlane->sgpr");
    run<2>(out, "48 VALU + 8 ballot+s_branch");
    run<3>(out, "56 VALU + 8 exec-masked ifs");
    return 0;
}
