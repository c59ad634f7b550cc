// Micro-benchmark: issue cost of the VALU instructions the MPI kernels use, on gfx950.
// Each kernel runs a long unrolled stream of ONE instruction on 8 independent registers per wave (no dependency stalls),
// 256 CUs x 4 SIMDs x W waves.  cycles/instr/SIMD = time * clock / (instrs per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define REP 64
#define ITERS 256
#define OPK(NAME, ASM)                                                                                      \
    __global__ void __launch_bounds__(256) k_##NAME(float *out, float seed)                                 \
    {                                                                                                       \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = seed * 0.5f + 1.0f, c = seed + 2.0f;                                                      \
        for (int it = 0; it < ITERS; ++it) {                                                                \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) {                                           \
                asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%4) "\n" ASM(%5) "\n" ASM(%6) "\n" ASM(%7) \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                             : "v"(b), "v"(c));                                                             \
            }                                                                                               \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                        \
    }
#define A_ADD(x) "v_add_f32 " #x ", " #x ", %8"
#define A_MUL(x) "v_mul_f32 " #x ", " #x ", %8"
#define A_FMA(x) "v_fma_f32 " #x ", " #x ", %8, %9"
#define A_FMAC(x) "v_fmac_f32 " #x ", %8, %9"
#define A_MAX(x) "v_max_f32 " #x ", " #x ", %8"
#define A_FLOOR(x) "v_floor_f32 " #x ", " #x
#define A_RNDNE(x) "v_rndne_f32 " #x ", " #x
#define A_CVTI(x) "v_cvt_i32_f32 " #x ", " #x
#define A_CVTF(x) "v_cvt_f32_i32 " #x ", " #x
#define A_RCP(x) "v_rcp_f32 " #x ", " #x
#define A_SQRT(x) "v_sqrt_f32 " #x ", " #x
#define A_LDEXP(x) "v_ldexp_f32 " #x ", " #x ", %8"
#define A_MED3(x) "v_med3_f32 " #x ", " #x ", %8, %9"
#define A_ADDU(x) "v_add_u32 " #x ", " #x ", %8"
#define A_MULU24(x) "v_mul_u32_u24 " #x ", " #x ", %8"
#define A_MULLO(x) "v_mul_lo_u32 " #x ", " #x ", %8"
#define A_LSHLADD(x) "v_lshl_add_u32 " #x ", " #x ", 4, %8"
#define A_CMPCND(x) "v_cmp_lt_f32 vcc, " #x ", %8\n v_cndmask_b32 " #x ", " #x ", %9, vcc"
#define A_CND(x) "v_cndmask_b32 " #x ", " #x ", %9, vcc"
#define A_CMP(x) "v_cmp_lt_f32 vcc, " #x ", %8"
#define A_MOV(x) "v_mov_b32 " #x ", %8"
#define A_PKFMA(x) "v_pk_fma_f32 " #x ", " #x ", %8, %9"   /* placeholder: needs 64-bit regs, handled separately */
OPK(add, A_ADD) OPK(mul, A_MUL) OPK(fma, A_FMA) OPK(fmac, A_FMAC) OPK(max, A_MAX) OPK(floor, A_FLOOR) OPK(rndne, A_RNDNE)
OPK(cvti, A_CVTI) OPK(cvtf, A_CVTF) OPK(rcp, A_RCP) OPK(sqrt, A_SQRT) OPK(ldexp, A_LDEXP) OPK(med3, A_MED3) OPK(addu, A_ADDU)
OPK(mulu24, A_MULU24) OPK(mullo, A_MULLO) OPK(lshladd, A_LSHLADD) OPK(cmpcnd, A_CMPCND) OPK(cnd, A_CND) OPK(cmp, A_CMP) OPK(mov, A_MOV)

// fp64 ops on 4 independent doubles
#define OPK64(NAME, ASM)                                                                                    \
    __global__ void __launch_bounds__(256) k_##NAME(float *out, float seed)                                 \
    {                                                                                                       \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                              \
        double b = seed * 0.5 + 1.0;                                                                        \
        for (int it = 0; it < ITERS; ++it) {                                                                \
            _Pragma("unroll") for (int r = 0; r < REP / 4; ++r) {                                           \
                asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b)); \
            }                                                                                               \
        }                                                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3);                                   \
    }
#define A_MUL64(x) "v_mul_f64 " #x ", " #x ", %4"
#define A_ADD64(x) "v_add_f64 " #x ", " #x ", %4"
#define A_PKFMA64(x) "v_pk_fma_f32 " #x ", " #x ", %4, %4"
#define A_PKMUL64(x) "v_pk_mul_f32 " #x ", " #x ", %4"
#define A_PKADD64(x) "v_pk_add_f32 " #x ", " #x ", %4"
OPK64(mul64, A_MUL64) OPK64(add64, A_ADD64) OPK64(pkfma, A_PKFMA64) OPK64(pkmul, A_PKMUL64) OPK64(pkadd, A_PKADD64)

__global__ void __launch_bounds__(256) k_cvt64(float *out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double d0, d1, d2, d3;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n"
                         "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <typename K> static double run(K kern, float *d, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0 * 1e-3;
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    const int wps = 4;                       // waves per SIMD: 256 CUs * 4 WG/CU (each WG = 4 waves, 1 per SIMD)
    const int blocks = 256 * wps;
    const double clk = 2.4e9;
    const double n = (double)ITERS * REP * wps;   // instructions per SIMD
    printf("cycles per wave64 instruction per SIMD at %.1f GHz nominal (lower clock => overestimate), %d waves/SIMD\n", clk / 1e9, wps);
#define R(NAME, MUL) printf("%-10s %6.2f\n", #NAME, run(k_##NAME, d, blocks) * clk / (n * MUL));
    R(add, 1) R(mul, 1) R(fma, 1) R(fmac, 1) R(max, 1) R(mov, 1) R(floor, 1) R(rndne, 1) R(cvti, 1) R(cvtf, 1) R(rcp, 1) R(sqrt, 1) R(ldexp, 1) R(med3, 1)
    R(addu, 1) R(mulu24, 1) R(mullo, 1) R(lshladd, 1) R(cmp, 1) R(cnd, 1) R(cmpcnd, 2) R(mul64, 1) R(add64, 1) R(pkfma, 1) R(pkmul, 1) R(pkadd, 1) R(cvt64, 1)
    return 0;
}
