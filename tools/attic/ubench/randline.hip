// Microbenchmark: wave-uniform dependent random 64-byte line fetches (the packet-traversal access
// pattern) vs footprint, resident waves and loads in flight per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MLP independent chains per wave; each step: every chain loads one 64-B line (lane&15 dwords), next index
// depends on the loaded data.
template <int MLP>
__global__ __launch_bounds__(256) void chase(const uint32_t* __restrict__ buf, uint32_t n_lines, int steps, uint32_t* out, uint32_t window, uint32_t hot_lines, uint32_t hot_pct)
{
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t idx[MLP];
    for (int m = 0; m < MLP; m++) idx[m] = mix(wave * 977u + m * 131071u) % n_lines;
    uint32_t acc = 0;
    const uint32_t base0 = mix(wave) % n_lines;
    for (int s = 0; s < steps; s++) {
        uint32_t v[MLP];
#pragma unroll
        for (int m = 0; m < MLP; m++) v[m] = buf[(size_t)idx[m] * 16 + (lane & 15)];
#pragma unroll
        for (int m = 0; m < MLP; m++) {
            const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)v[m], 3);
            acc += r;
            uint32_t nx = mix(r + idx[m] + s);
            // window > 0: stay within a window of lines around a per-wave base (locality)
            idx[m] = window ? (base0 + nx % window) % n_lines : nx % n_lines;
            if (hot_lines && (nx >> 8) % 100u < hot_pct) idx[m] = (nx >> 16) % hot_lines;   // shared hot set
        }
    }
    if (lane == 0) out[wave] = acc;
}

int main()
{
    const size_t max_bytes = (size_t)4 << 30;
    uint32_t* buf; uint32_t* out;
    CK(hipMalloc(&buf, max_bytes));
    CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(buf, 1, max_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 2000;
    const uint32_t n_lines = (uint32_t)(((size_t)64 << 20) / 64);
    for (uint32_t hot_lines : {0u, 1u, 16u, 1024u}) {
        for (uint32_t hot_pct : {10u, 25u, 50u}) {
            if (hot_lines == 0 && hot_pct != 10u) continue;
            for (int blocks_per_cu : {1, 8}) {
                const int blocks = 256 * blocks_per_cu;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, 0, buf, n_lines, steps, out, 0u, hot_lines, hot_pct);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double lines = (double)blocks * 4 * steps;
                printf("hot_lines=%5u hot_pct=%3u waves=%5d  %8.3f ms %8.3f Glines/s %8.1f ns/step\n", hot_lines, hot_pct, blocks * 4, ms, lines / ms / 1e6, ms * 1e6 / steps);
            }
        }
    }
    return 0;
}
