// Which XCD does workgroup b of a launch run on?  (XCC_ID hardware register of every block's first wave)
// hipcc --offload-arch=gfx950 -O2 tools/ubench/xcdmap.hip -o tools/ubench/xcdmap && tools/ubench/xcdmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, int spin)
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v & 0xF;
    for (int i = 0; i < spin; i++) asm volatile("s_sleep 4");
}
int main()
{
    for (int threads : {64, 256, 512}) for (int spin : {0, 2000}) {
        const int blocks = 4096;
        unsigned* d; hipMalloc(&d, blocks * 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, spin);
        std::vector<unsigned> h(blocks); hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
        printf("threads %d spin %d first 32:", threads, spin);
        for (int b = 0; b < 32; b++) printf(" %u", h[b]);
        int same = 0; for (int b = 8; b < blocks; b++) same += h[b] == h[b % 8];
        printf("   blocks whose XCD == XCD of block b %% 8: %d / %d\n", same, blocks - 8);
        hipFree(d);
    }
    return 0;
}
