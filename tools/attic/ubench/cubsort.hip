// cubsort.hip — external sanity line for the radix sort: hipCUB / rocPRIM DeviceRadixSort::SortPairs on the
// same kind of input as bench.py's sort micro-bench (2^k uniform random u32 keys, values = iota).  Not part of the
// product; built by hand:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/cubsort tools/ubench/cubsort.hip
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

static uint32_t mix(uint32_t v)
{
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}

__global__ void fill(uint32_t* k, uint32_t* v, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = i * 2654435761u + 3u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    k[i] = x;
    v[i] = i;
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 26;
    const uint32_t n = 1u << log2n;
    uint32_t *k0, *k1, *v0, *v1;
    hipMalloc(&k0, n * 4ull); hipMalloc(&k1, n * 4ull); hipMalloc(&v0, n * 4ull); hipMalloc(&v1, n * 4ull);
    size_t tmp_bytes = 0;
    hipcub::DoubleBuffer<uint32_t> dk(k0, k1), dv(v0, v1);
    hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dk, dv, (int)n);
    void* tmp = nullptr;
    hipMalloc(&tmp, tmp_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f, sum = 0.0f;
    const int reps = 6;
    for (int r = 0; r < reps; r++) {
        hipcub::DoubleBuffer<uint32_t> kk(k0, k1), vv(v0, v1);
        fill<<<(n + 255) / 256, 256>>>(k0, v0, n);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kk, vv, (int)n);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, a, b);
        if (r > 0) { sum += ms; if (ms < best) best = ms; }
        if (r == reps - 1) {        // sortedness of the result
            std::vector<uint32_t> h(1u << 20);
            hipMemcpy(h.data(), kk.Current(), h.size() * 4, hipMemcpyDeviceToHost);
            bool ok = true;
            for (size_t i = 1; i < h.size(); i++) ok = ok && h[i - 1] <= h[i];
            printf("sorted prefix ok: %d\n", (int)ok);
        }
    }
    const float mean = sum / (reps - 1);
    printf("{\"hipcub_sort_pairs\": {\"log2n\": %d, \"ms_mean\": %.4f, \"ms_best\": %.4f, \"Gkeys_s\": %.2f, \"temp_MB\": %.1f}}\n", log2n, mean,
           best, n / (mean * 1e-3) / 1e9, tmp_bytes / 1048576.0);
    (void)mix;
    return 0;
}
