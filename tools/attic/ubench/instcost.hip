// Per-instruction issue cost on gfx950 at 1 .. 8 waves per SIMD: 64 copies of ONE instruction kind per loop iteration
// (one inline-asm block: explicit registers, no compiler-inserted s_nop), independent destinations unless the name says
// otherwise.  Prints cycles per instruction per SIMD from wall time and the in-kernel clock (s_memtime over
// s_memrealtime), counting only the share of the launch the waves were alive.  Feeds DESIGN.md section 7's ceiling.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define R8(a, b) a "0" b a "1" b a "2" b a "3" b a "4" b a "5" b a "6" b a "7" b
// 64 instructions: destination v[10 + k % 8] (or s[20 + 2 (k % 8)]) so that consecutive ones are independent
#define X64(pre, post) R8(pre, post) R8(pre, post) R8(pre, post) R8(pre, post) R8(pre, post) R8(pre, post) R8(pre, post) R8(pre, post)
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
             "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "vcc", "scc"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed)
{
    float a = threadIdx.x * 0.001f + seed;
    asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n"
                 "v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n"
                 "v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n s_mov_b32 s36, 1\n s_mov_b32 s37, 2"
                 :: "v"(a) : CLOB);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(X64("v_mul_f32 v1", ", v20, v21\n") ::: CLOB);
        if (MODE == 1) asm volatile(X64("v_fma_f32 v1", ", v20, v21, v22\n") ::: CLOB);                   // banks 0, 1, 2
        if (MODE == 2) asm volatile(X64("v_fma_f32 v1", ", v20, v24, v20\n") ::: CLOB);                   // all sources in bank 0
        if (MODE == 3) asm volatile(X64("v_sub_f32_dpp v1", ", v20, v21 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") ::: CLOB);
        if (MODE == 4) asm volatile(X64("v_max3_f32 v1", ", v20, v21, v22\n") ::: CLOB);
        if (MODE == 5) asm volatile(X64("v_cmp_gt_f32 vcc, v2", ", v21\n") ::: CLOB);
        if (MODE == 6) asm volatile(X64("v_cmp_gt_f32 s[38:39], v2", ", v21\n") ::: CLOB);
        if (MODE == 7) asm volatile(X64("v_readlane_b32 s4", ", v21, 5\n") ::: CLOB);
        if (MODE == 8) asm volatile(X64("v_cndmask_b32 v1", ", v20, v21, vcc\n") ::: CLOB);
        if (MODE == 9) asm volatile(X64("v_mov_b32 v1", ", v20\n") ::: CLOB);
        if (MODE == 10) asm volatile(X64("s_add_u32 s4", ", s36, s37\n") ::: CLOB);
        if (MODE == 11) asm volatile(X64("s_and_b64 s[38:39], s[36:37], vcc ; ", "\n") ::: CLOB);
        if (MODE == 12) asm volatile(X64("v_mul_f32 v1", ", v20, v21\n s_add_u32 s40, s36, s37 ; ") ::: CLOB);   // 64 VALU + 64 SALU interleaved
        if (MODE == 13) asm volatile(X64("v_mul_f32 v10, v10, v21 ; ", "\n") ::: CLOB);                  // one dependent chain
        if (MODE == 14) asm volatile(X64("v_min_f32 v1", ", v20, v21\n") ::: CLOB);
        if (MODE == 15) asm volatile(X64("v_mul_f32 v1", ", s36, v21\n") ::: CLOB);                      // one SGPR operand
        if (MODE == 16) asm volatile(X64("v_writelane_b32 v1", ", s36, 3\n") ::: CLOB);
        if (MODE == 17) asm volatile(X64("s_bcnt1_i32_b64 s4", ", s[36:37]\n") ::: CLOB);
        if (MODE == 18) asm volatile(X64("v_cndmask_b32_e64 v1", ", v20, v21, s[36:37]\n") ::: CLOB);
        if (MODE == 19) asm volatile(X64("v_add_f32 v1", ", v20, v21\n") ::: CLOB);
        if (MODE == 20) asm volatile(X64("v_sub_f32 v1", ", v20, v21\n") ::: CLOB);
        if (MODE == 21) asm volatile(X64("v_max_f32 v1", ", v20, v21\n") ::: CLOB);
        if (MODE == 22) asm volatile(X64("v_pk_mul_f32 v[10:11], v[20:21], v[22:23] ; ", "\n") ::: CLOB);
        if (MODE == 23) asm volatile(X64("v_pk_add_f32 v[10:11], v[20:21], v[22:23] ; ", "\n") ::: CLOB);
        if (MODE == 24) asm volatile(X64("v_and_b32 v1", ", v20, v21\n") ::: CLOB);
        if (MODE == 25) asm volatile(X64("v_cmp_lt_u32 vcc, v2", ", v21\n") ::: CLOB);
        if (MODE == 26) asm volatile(X64("v_mov_b32_dpp v1", ", v20 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") ::: CLOB);
        if (MODE == 27) asm volatile(X64("v_fmac_f32 v1", ", v20, v21\n") ::: CLOB);
        if (MODE == 28) asm volatile(X64("v_fmac_f32_dpp v1", ", v20, v21 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") ::: CLOB);
        if (MODE == 29) asm volatile(X64("v_add_u32 v1", ", v20, v21\n") ::: CLOB);
        if (MODE == 30) asm volatile(X64("v_mul_f32 v1", ", v20, v20\n") ::: CLOB);                     // same register twice
        if (MODE == 31) asm volatile(X64("v_mul_f32 v1", ", v20, v24\n") ::: CLOB);                     // two registers of one bank
        if (MODE == 32) asm volatile(X64("v_min3_f32 v1", ", v20, v21, v22\n") ::: CLOB);
        if (MODE == 33) asm volatile(X64("v_med3_f32 v1", ", v20, v21, v22\n") ::: CLOB);
        if (MODE == 34) asm volatile(X64("v_mul_f32 v1", ", 2.0, v21\n") ::: CLOB);                     // inline constant
        if (MODE == 35) asm volatile(X64("v_mul_f32_e64 v1", ", v20, v21 clamp\n") ::: CLOB);           // VOP3 encoding of a VOP2 op
        if (MODE == 36) asm volatile(X64("v_cmp_gt_f32 vcc, v2", ", v21\n s_and_b64 s[38:39], vcc, s[36:37] ; ") ::: CLOB);
        if (MODE == 38) asm volatile(X64("v_cndmask_b32_e64 v1", ", v20, v21, vcc\n") ::: CLOB);
        if (MODE == 39) asm volatile(X64("v_cmp_gt_f32 vcc, v2", ", v21\n v_cndmask_b32 v10, v20, v21, vcc ; ") ::: CLOB);
        if (MODE == 40) asm volatile(X64("v_cmp_gt_f32 s[38:39], v2", ", v21\n v_cndmask_b32_e64 v10, v20, v21, s[38:39] ; ") ::: CLOB);
        if (MODE == 41) asm volatile("s_mov_b64 vcc, 0x55\n" X64("v_cndmask_b32 v1", ", v20, v21, vcc\n") ::: CLOB);
        if (MODE == 42) asm volatile(X64("v_cndmask_b32 v1", ", v20, v20, vcc\n") ::: CLOB);
        if (MODE == 43) asm volatile(X64("v_addc_co_u32 v1", ", vcc, v20, v21, vcc\n") ::: CLOB);
        if (MODE == 44) asm volatile(X64("v_cndmask_b32 v1", ", 0, v21, vcc\n") ::: CLOB);
        if (MODE == 37) asm volatile(X64("v_max_f32 v1", ", v20, v20\n") ::: CLOB);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float r;
    asm volatile("v_add_f32 %0, v10, v11\n v_add_f32 %0, %0, v12" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
        cyc[256 * 8 * 4 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = r1 - r0;
    }
}

template <int MODE>
void run(float* out, unsigned long long* cyc, const char* name, int per_iter = 64)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 512;
    static unsigned long long h[256 * 8 * 4 * 2];
    printf("%-44s", name);
    for (int bpc : {1, 2, 4, 8}) {
        const int blocks = 256 * bpc;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.5f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        double mean = 0, real = 0;
        for (int i = 0; i < blocks * 4; i++) { mean += (double)h[i]; real += (double)h[256 * 8 * 4 + i]; }
        mean /= blocks * 4; real /= blocks * 4;
        const double clock_ghz = mean / (real * 10.0);
        const double insts = (double)iters * per_iter;
        // cycles the SIMD spends per instruction: launch wall time x clock / (instructions one SIMD issued)
        printf("  w%d: %5.2f cyc/wave %5.2f cyc/SIMD (%.2f GHz)", bpc, mean / insts, ms * 1e-3 * clock_ghz * 1e9 / (insts * bpc), clock_ghz);
    }
    printf("\n");
}

int main()
{
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 256 * 8 * 4 * 8 * 2));
    printf("cycles per instruction: per wave (s_memtime) and per SIMD (launch time x clock / instructions per SIMD), 1 / 2 / 4 / 8 workgroups of 4 waves per CU\n");
    run<0>(out, cyc, "v_mul_f32 (VOP2)");
    run<14>(out, cyc, "v_min_f32 (VOP2)");
    run<15>(out, cyc, "v_mul_f32 with an SGPR operand");
    run<1>(out, cyc, "v_fma_f32, sources in 3 banks");
    run<2>(out, cyc, "v_fma_f32, sources in 1 bank");
    run<13>(out, cyc, "v_mul_f32, one dependent chain");
    run<3>(out, cyc, "v_sub_f32_dpp row_newbcast");
    run<4>(out, cyc, "v_max3_f32");
    run<5>(out, cyc, "v_cmp_gt_f32 -> vcc");
    run<6>(out, cyc, "v_cmp_gt_f32 -> sgpr pair");
    run<7>(out, cyc, "v_readlane_b32 -> sgpr");
    run<16>(out, cyc, "v_writelane_b32");
    run<8>(out, cyc, "v_cndmask_b32 (vcc)");
    run<9>(out, cyc, "v_mov_b32");
    run<18>(out, cyc, "v_cndmask_b32_e64 (sgpr pair)");
    run<38>(out, cyc, "v_cndmask_b32_e64 (vcc named)");
    run<41>(out, cyc, "v_cndmask_b32 e32, vcc set by s_mov first");
    run<42>(out, cyc, "v_cndmask_b32 e32, both sources v20");
    run<44>(out, cyc, "v_cndmask_b32 e32, src0 = 0");
    run<39>(out, cyc, "v_cmp->vcc + v_cndmask e32 (per pair)");
    run<40>(out, cyc, "v_cmp->sgpr + v_cndmask e64 (per pair)");
    run<43>(out, cyc, "v_addc_co_u32 (vcc in and out)");
    run<19>(out, cyc, "v_add_f32");
    run<20>(out, cyc, "v_sub_f32");
    run<21>(out, cyc, "v_max_f32");
    run<37>(out, cyc, "v_max_f32 v, v20, v20");
    run<32>(out, cyc, "v_min3_f32");
    run<33>(out, cyc, "v_med3_f32");
    run<22>(out, cyc, "v_pk_mul_f32 (WAW on one pair)");
    run<23>(out, cyc, "v_pk_add_f32 (WAW on one pair)");
    run<24>(out, cyc, "v_and_b32");
    run<29>(out, cyc, "v_add_u32");
    run<25>(out, cyc, "v_cmp_lt_u32 -> vcc");
    run<26>(out, cyc, "v_mov_b32_dpp row_newbcast");
    run<27>(out, cyc, "v_fmac_f32");
    run<28>(out, cyc, "v_fmac_f32_dpp row_newbcast");
    run<30>(out, cyc, "v_mul_f32 v, v20, v20");
    run<31>(out, cyc, "v_mul_f32 v, v20, v24 (one bank)");
    run<34>(out, cyc, "v_mul_f32 v, 2.0, v21");
    run<35>(out, cyc, "v_mul_f32_e64 clamp (VOP3)");
    run<36>(out, cyc, "v_cmp_gt_f32 vcc + s_and_b64 (per pair)");
    run<10>(out, cyc, "s_add_u32");
    run<11>(out, cyc, "s_and_b64");
    run<17>(out, cyc, "s_bcnt1_i32_b64");
    run<12>(out, cyc, "v_mul_f32 + s_add_u32 interleaved (per pair)");
    return 0;
}
