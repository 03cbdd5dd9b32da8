// Issue-throughput microbenchmark: how many cycles does one wave64 vector instruction cost a
// SIMD on gfx950 with 1 / 2 / 4 / 8 resident waves per SIMD, for the instruction mixes the packet traversal runs?
// No memory traffic in the loops.  Cycles are the shader clock itself: every wave brackets its loop with s_memtime
// and the table prints (mean cycles per wave) x 1 / (vector instructions per wave) / ... per SIMD:
//     cyc/inst/SIMD = wave_cycles / (insts_per_wave * waves_per_SIMD)
// MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on the SIMD-32 when other waves are resident; one wave alone: 4.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// VALU instructions per loop iteration of each mode (checked against the ISA: tools/ubench/issue.s)
static const int kValuPerIter[8] = {64, 64, 80, 79, 64, 64, 64, 64};   // as compiled with -O3 -fno-slp-vectorize (counted in the ISA)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed)
{
    float a = threadIdx.x * 0.001f + seed, b = a * 1.5f, c = b + 2.0f, d = c * 0.7f;
    float e = a + 3.0f, f = b + 5.0f, g = c + 7.0f, h = d + 11.0f;
    int acc = 0;
    const float sv = seed, sw = seed + 1.0f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a, b}, p1 = {c, d}, p2 = {e, f}, p3 = {g, h}; const f2 pw = {seed, seed};
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 64 VALU, VGPR operands only, partly dependent (4 accumulators in a ring)
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
        } else if (MODE == 3) {   // 48 VALU + 8 exec-masked regions (s_and_saveexec / s_or exec / cbranch_execz)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                a = a * b + c; b = b * c + d; c = c * d + a; d = d * a + b; a = a * b + d; b = b + c;
                if (a > b + (float)j) { c = c * 1.0001f + d; }
            }
        } else if (MODE == 4) {   // 64 VALU, 8 INDEPENDENT accumulators: no instruction waits for its predecessor
#pragma unroll
            for (int j = 0; j < 8; j++) {
                a = a * sv + 1.0f; b = b * sv + 1.0f; c = c * sv + 1.0f; d = d * sv + 1.0f;
                e = e * sv + 1.0f; f = f * sv + 1.0f; g = g * sv + 1.0f; h = h * sv + 1.0f;
            }
        } else if (MODE == 5) {   // 64 VALU in ONE dependent chain: every instruction needs the one before it
#pragma unroll
            for (int j = 0; j < 64; j++) a = a * sv + 1.0f;
        } else if (MODE == 6) {   // the box-test mix, independent: 24 sub(DPP)/mul, 4 min3/max3, 4 cmp per 32
#pragma unroll
            for (int j = 0; j < 2; j++) {
                asm volatile(
                    "v_sub_f32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                    "v_sub_f32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                    "v_sub_f32_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                    "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n"
                    "v_max3_f32 %6, %0, %1, %2\n v_min3_f32 %7, %3, %4, %5\n v_cmp_gt_f32 vcc, %7, %6\n v_cmp_gt_f32 vcc, %7, %9\n"
                    "v_sub_f32_dpp %0, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %1, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
                    "v_sub_f32_dpp %2, %8, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %3, %8, %9 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
                    "v_sub_f32_dpp %4, %8, %9 row_newbcast:13 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %5, %8, %9 row_newbcast:14 row_mask:0xf bank_mask:0xf\n"
                    "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n"
                    "v_max3_f32 %6, %0, %1, %2\n v_min3_f32 %7, %3, %4, %5\n v_cmp_gt_f32 vcc, %7, %6\n v_cmp_gt_f32 vcc, %7, %9"
                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(sv), "v"(sw) : "vcc");
            }
        } else {   // MODE 7: 64 v_pk_fma_f32, 8 independent accumulators pairs (does the packed form issue at the same rate?)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pw));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + acc + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
        cyc[256 * 8 * 4 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = r1 - r0;       // 100 MHz ticks
    }
}

template <int MODE>
void run(float* out, unsigned long long* cyc, const char* name, int iters = 20000, float seed = 0.5f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    static unsigned long long h[256 * 8 * 4 * 2];
    for (int bpc : {1, 2, 3, 4, 6, 8}) {
        const int blocks = 256 * bpc;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, seed);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, cyc, sizeof(unsigned long long) * 256 * 8 * 4 * 2, hipMemcpyDeviceToHost));
        double mean = 0, real = 0;
        for (int i = 0; i < blocks * 4; i++) { mean += (double)h[i]; real += (double)h[256 * 8 * 4 + i]; }
        mean /= blocks * 4; real /= blocks * 4;     // real: mean wave lifetime in 100 MHz ticks
        const double clock_ghz = mean / (real * 10.0), resident = real * 1e-5 / ms;   // share of the launch a wave is alive
        const double insts = (double)iters * kValuPerIter[MODE];
        printf("%-30s iters=%-5d waves/SIMD=%d  %8.3f ms  wave cycles %9.0f  clock %.3f GHz  wave alive %.2f of launch  cyc/VALU/wave %6.2f  cyc/VALU/SIMD(all waves alive) %5.2f  G wave-insts/s chip %7.1f\n",
               name, iters, bpc, ms, mean, clock_ghz, resident, mean / insts, mean / (insts * bpc), insts * blocks * 4 / (ms * 1e6));
    }
}

int main()
{
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 256 * 8 * 4 * 8 * 2));
    // short launches first (0.05 - 0.5 ms: the traversal kernel's time scale, before the power management reacts)
    printf("---- short launches (512 iterations)\n");
    run<4>(out, cyc, "64 v_fma independent (8 acc)", 512);
    run<6>(out, cyc, "64 box-test mix (dpp,min3,cmp)", 512);
    run<0>(out, cyc, "64 VALU ring of 4", 512);
    run<2>(out, cyc, "48 VALU+8 ballot+s_branch", 512);
    run<1>(out, cyc, "48 VALU+16 readlane->sgpr", 512);
    printf("---- the same instruction stream on infinities (operands that do not toggle the datapath)\n");
    run<4>(out, cyc, "64 v_fma independent, inf data", 512, __builtin_inff());
    run<4>(out, cyc, "64 v_fma independent, inf data", 20000, __builtin_inff());
    printf("---- long launches (20000 iterations: 3 - 18 ms, the clock comes down under load)\n");
    run<4>(out, cyc, "64 v_fma independent (8 acc)");
    run<5>(out, cyc, "64 v_fma one dependent chain");
    run<6>(out, cyc, "64 box-test mix (dpp,min3,cmp)");
    run<7>(out, cyc, "64 v_pk_fma_f32 independent");
    run<0>(out, cyc, "64 VALU ring of 4");
    run<1>(out, cyc, "48 VALU+16 readlane->sgpr");
    run<2>(out, cyc, "48 VALU+8 ballot+s_branch");
    run<3>(out, cyc, "56 VALU+8 exec-masked ifs");
    return 0;
}
