// rocprim_sort.hip — an EXTERNALLY written stable pair sort on the GPU box, as a third checker of the sort's contract
// (VERDICT r2 item 8a).  TEST INFRASTRUCTURE ONLY, like everything under oracle/: tests/test_gpu_parity.py loads
// oracle/librocprim_sort.so and compares lbvh_sort_pairs' keys AND values with rocPRIM's DeviceRadixSort::SortPairs
// (through hipCUB, headers shipped with ROCm under /opt/rocm/include) on the same input.  Both implement "stable
// sort of (key, value) pairs by key", which has exactly one answer — the reference's ComputeBufferSorter.Sort contract
// (Assets/_Scripts/ComputeBufferSorter.cs:100-126).  The product never links or loads this file.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

extern "C" {

// host arrays in, host arrays out (in place); returns 0 on success, the HIP error code otherwise
int rocprim_sort_pairs(uint32_t* h_keys, uint32_t* h_values, uint32_t count)
{
    if (count == 0) return 0;
    uint32_t *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr;
    void* tmp = nullptr;
    const size_t bytes = (size_t)count * 4;
    hipError_t e = hipMalloc(&k0, bytes);
    if (e == hipSuccess) e = hipMalloc(&k1, bytes);
    if (e == hipSuccess) e = hipMalloc(&v0, bytes);
    if (e == hipSuccess) e = hipMalloc(&v1, bytes);
    if (e == hipSuccess) e = hipMemcpy(k0, h_keys, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v0, h_values, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipcub::DoubleBuffer<uint32_t> dk(k0, k1), dv(v0, v1);
        size_t tmp_bytes = 0;
        e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dk, dv, (int)count);
        if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
        if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, dk, dv, (int)count);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(h_keys, dk.Current(), bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(h_values, dv.Current(), bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(k0); (void)hipFree(k1); (void)hipFree(v0); (void)hipFree(v1); (void)hipFree(tmp);
    return (int)e;
}

}  // extern "C"
