// Copy-bandwidth microbenchmark: which float4 copy shape reaches the box's best HBM rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copyk(float4* __restrict__ dst_, const float4* __restrict__ src_, size_t n16)
{
    f4* __restrict__ dst = reinterpret_cast<f4*>(dst_);
    const f4* __restrict__ src = reinterpret_cast<const f4*>(src_);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// one contiguous chunk per block (block-contiguous instead of grid-strided)
template <int UNROLL>
__global__ __launch_bounds__(256) void copyc(float4* __restrict__ dst, const float4* __restrict__ src, size_t n16)
{
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
    for (size_t i = b0 + threadIdx.x; i < b1; i += 256 * UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) if (i + u * 256 < b1) v[u] = src[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) if (i + u * 256 < b1) dst[i + u * 256] = v[u];
    }
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    float4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n16 = bytes / 16;
    auto run = [&](const char* name, auto launch) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-40s %8.3f ms  %7.1f GB/s (read+write)\n", name, best, 2.0 * bytes / best / 1e6);
    };
    for (int blocks : {2048, 4096, 16384, 65536}) {
        char nm[64];
        snprintf(nm, 64, "grid-stride u4 blocks=%d", blocks); run(nm, [&] { hipLaunchKernelGGL((copyk<4, false>), dim3(blocks), dim3(256), 0, 0, b, a, n16); });
        snprintf(nm, 64, "grid-stride u8 blocks=%d", blocks); run(nm, [&] { hipLaunchKernelGGL((copyk<8, false>), dim3(blocks), dim3(256), 0, 0, b, a, n16); });
        snprintf(nm, 64, "grid-stride u4 nt blocks=%d", blocks); run(nm, [&] { hipLaunchKernelGGL((copyk<4, true>), dim3(blocks), dim3(256), 0, 0, b, a, n16); });
        snprintf(nm, 64, "block-chunk u4 blocks=%d", blocks); run(nm, [&] { hipLaunchKernelGGL((copyc<4>), dim3(blocks), dim3(256), 0, 0, b, a, n16); });
    }
    run("one float4 per thread (no loop)", [&] { hipLaunchKernelGGL((copyk<1, false>), dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, b, a, n16); });
    run("hipMemcpyDtoD", [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
