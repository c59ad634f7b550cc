// mpf_pconv.hip - the PARITY-GRADE engine of the MPI producer network: every convolution of
// MPIPredictor.forward (reference model/AdaMPI.py:55-78) - the RGBD ResNet-18 encoder (model/CPN/encoder.py:20-101), the
// feature-mask UNet (model/CPN/unet.py:18-69) and the gated decoder (model/CPN/decoder.py:10-71, :124-174) - in the arithmetic
// of the reference's CPU path: fp32 storage, fp32 products, fp32 accumulation (v_mfma_f32_16x16x4_f32) in blocks of 64 products
// whose partial sums are carried in fp64, or - same code, T = double - fp64 throughout (v_mfma_f64_16x16x4_f64), which is what
// the tests use to show that the engine computes the reference's network and not something 1e-4 away from it.
//
// A third form, what --model-dtype fp32 runs: the same fp32 tensors and the same two-level sum with every PRODUCT taken from the three bf16 numbers each fp32 factor
// is exactly the sum of, on v_mfma_f32_16x16x32_bf16 - the matrix cores instead of the vector-rate fp32 instruction (k_pconv_x3, k_pconv_x3_tile, k_pconv_x3_chunk
// below: 37 instead of 59 ms per image and three times closer to exact arithmetic).
//
// This is the accuracy mode, not the fast one (that is mpf_conv.hip: fp16 storage, 7.7 ms per image).  Inputs are plain
// materialised NHWC tensors; the only synthesis left in the loader is what costs nothing: the concatenation of two sources,
// the x2 nearest up-sampling of the first and reflection / zero padding.
//
// k_pconv: implicit GEMM  out[row, pixel] = sum_{tap, c} W[row, tap, c] * in[pixel + tap, c].  A wave owns NB 16-row blocks
// of output rows x PG groups of 16 consecutive pixels of one plane and runs the whole K loop for them (no split-K, no LDS, no
// barrier: the sum order is fixed by construction).  K = source A's (tap, 4-channel vector) pairs, then source B's, four vectors per
// K-step: lane (m = l % 16, g = l / 16) loads the 4 channels of K-vector 4 step + g at pixel m (one 16- or 32-byte buffer load) and the
// matching 4 weights of row m; the 4 elements feed 4 MFMAs (any permutation of K is a valid GEMM as long as both operands use it).
#include "mpf_common.h"
#include <type_traits>
#ifndef MPF_X3_ABLATE
#define MPF_X3_ABLATE 0
#endif

namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<double> { typedef double type __attribute__((ext_vector_type(4))); };
template <typename T> struct Vec2;
template <> struct Vec2<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec2<double> { typedef double type __attribute__((ext_vector_type(2))); };

__device__ __forceinline__ Vec4<float>::type mfma16(float a, float b, Vec4<float>::type c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ Vec4<double>::type mfma16(double a, double b, Vec4<double>::type c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// one 4-channel vector through a buffer descriptor: a byte offset at or beyond num_records returns zeros - padding pixels need no select
__device__ __forceinline__ Vec4<float>::type buf_load4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float)
{
    return __builtin_bit_cast(Vec4<float>::type, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}
__device__ __forceinline__ Vec4<double>::type buf_load4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double)
{
    struct { u32x4_t lo, hi; } r = { __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0), __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16u, soff, 0) };
    return __builtin_bit_cast(Vec4<double>::type, r);
}

// correctly-rounded-grade library functions (ocml; the translation unit is built with -fno-fast-math)
__device__ __forceinline__ float exp_t(float x) { return expf(x); }
__device__ __forceinline__ double exp_t(double x) { return exp(x); }
__device__ __forceinline__ float expm1_t(float x) { return expm1f(x); }
__device__ __forceinline__ double expm1_t(double x) { return expm1(x); }
template <typename T> __device__ __forceinline__ T sigmoid_t(T x) { return (T)1 / ((T)1 + exp_t(-x)); }        // model/CPN/decoder.py:69-70

constexpr int EP_AFFINE = MPF_PCONV_EP_AFFINE, EP_AFFINE_MAP = MPF_PCONV_EP_AFFINE_MAP, EP_GATED = MPF_PCONV_EP_GATED, EP_GATED_PLANAR = MPF_PCONV_EP_GATED_PLANAR;

template <typename T>
__device__ __forceinline__ T act_apply(T y, int act, T slope)
{
    if (act == 1) return y > (T)0 ? y : (T)0;
    if (act == 2) return y > (T)0 ? y : y * slope;
    return y;
}

// The epilogue of every convolution kernel.  lane (m, g): acc[nb][pg][i] = LOGICAL row 4g + i of block bg * NB + nb at pixel pix[pg] of the plane (P pixels; a
// value >= P: no pixel).  The host permutes the rows of a block for the fp64 instruction, whose C/D layout is row = g + 4 i: mpiflow_amd/model/precise.py
template <typename T, int NB, int PG>
__device__ __forceinline__ void pconv_epilogue(const MpfPConvArgs &a, const typename Vec4<T>::type (&acc)[NB][PG], const int (&pix)[PG], const int g, const int bg,
                                               const int s, const int P)
{
    typedef typename Vec4<T>::type v4;
    typedef typename Vec2<T>::type v2;
    const T *scale = reinterpret_cast<const T *>(a.scale), *shift = reinterpret_cast<const T *>(a.shift);
    T *out = reinterpret_cast<T *>(a.out);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int p = pix[pg];
        if (p >= P) continue;
        const size_t opix = (size_t)s * P + p;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int blk = bg * NB + nb;
            const v4 c = acc[nb][pg];
            if (a.epi == EP_AFFINE || a.epi == EP_AFFINE_MAP) {
                const int ch = blk * 16 + 4 * g;
                const v4 sc = *reinterpret_cast<const v4 *>(scale + ch), sh = *reinterpret_cast<const v4 *>(shift + ch);
                v4 y;
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = c[i] * sc[i] + sh[i];       // eval-mode BatchNorm folded with the conv bias (two roundings, as written)
                if (a.epi == EP_AFFINE_MAP) {
                    if (ch == 0) out[opix] = act_apply<T>(y[0], a.act, (T)a.slope);       // single-channel map [S,H,W]
                } else if (ch < a.Cst) {
                    if (a.residual) {
                        const v4 r = *reinterpret_cast<const v4 *>(reinterpret_cast<const T *>(a.residual) + opix * a.Cst + ch);
#pragma unroll
                        for (int i = 0; i < 4; ++i) y[i] += r[i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = act_apply<T>(y[i], a.act, (T)a.slope);
                    *reinterpret_cast<v4 *>(out + opix * a.Cst + ch) = y;
                }
            } else {
                // logical rows (4g, 4g+1) = (feature, gate) of channel 8 blk + 2g, rows (4g+2, 4g+3) of channel 8 blk + 2g + 1; the biases were the
                // accumulators' initial values.  model/CPN/decoder.py:66-70: conv2d(x) * sigmoid(mask_conv2d(x))
                const int ch = blk * 8 + 2 * g;
                T y[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) y[j] = c[2 * j] * sigmoid_t<T>(c[2 * j + 1]);
                if (a.epi == EP_GATED) {                                // + BatchNorm + ELU (model/CPN/decoder.py:36-40)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const T t = y[j] * scale[ch + j] + shift[ch + j];
                        y[j] = t > (T)0 ? t : expm1_t(t);
                    }
                    if (ch < a.Cst) *reinterpret_cast<v2 *>(out + opix * a.Cst + ch) = v2{y[0], y[1]};
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (ch + j < a.Cst) out[((size_t)s * a.Cst + ch + j) * P + p] = y[j];   // planar [S, Cst, H, W]
                }
            }
        }
    }
}

template <typename T, int NB, int PG>
__global__ __launch_bounds__(256) void k_pconv(const MpfPConvArgs a)
{
    typedef typename Vec4<T>::type v4;
    typedef typename Vec2<T>::type v2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
    const int P = a.Hout * a.Wout;
    const int p0 = ((int)blockIdx.x * 4 + wave) * (16 * PG);
    if (p0 >= P) return;                                              // no barrier in this kernel: a wave without pixels may leave
    const int bg = blockIdx.y, s = blockIdx.z;
    const int VA = a.CA >> 2, VB = a.CB >> 2, ks = a.ksize;
    const int nstA = (ks * ks * VA + 3) >> 2, nstB = (ks * ks * VB + 3) >> 2, nsteps = nstA + nstB;      // K-steps of the two sources (K = source A's taps x vectors, then B's)
    int oy[PG], ox[PG];
    bool pv[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int p = p0 + 16 * pg + m;
        pv[pg] = p < P;
        const int pc = pv[pg] ? p : 0;
        oy[pg] = pc / a.Wout;
        ox[pg] = pc - oy[pg] * a.Wout;
        oy[pg] = oy[pg] * a.stride - a.pad;
        ox[pg] = ox[pg] * a.stride - a.pad;
    }
    const v4 zero = {(T)0, (T)0, (T)0, (T)0};
    // fp32: the MFMA accumulators carry FLUSH k-steps (64 products) only and are then added into fp64 carries - a sequential fp32 chain over the
    // whole K (up to 4644 here) has a relative error ~ eps sqrt(K), the two-level sum ~ eps sqrt(64): below the blocked / vectorised fp32 sums of
    // the reference's CPU convolutions (oneDNN) instead of 2-3x above them (profiles/r5/precise_engine_error.txt).  fp64: one level.
    constexpr bool TWO_LEVEL = sizeof(T) == 4;
    constexpr int FLUSH = 4;
    typedef typename Vec4<double>::type v4d;
    v4 acc[NB][PG];
    v4d carry[TWO_LEVEL ? NB : 1][TWO_LEVEL ? PG : 1];
    const bool gated = a.epi == EP_GATED || a.epi == EP_GATED_PLANAR;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        v4 init = zero;
        if (gated) init = *reinterpret_cast<const v4 *>(reinterpret_cast<const T *>(a.bias) + (bg * NB + nb) * 16 + 4 * g);   // conv biases of logical rows 4g..4g+3
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            if constexpr (TWO_LEVEL) {
                carry[nb][pg] = v4d{(double)init[0], (double)init[1], (double)init[2], (double)init[3]};
                acc[nb][pg] = zero;
            } else {
                acc[nb][pg] = init;
            }
        }
    }
    const bool reflect = a.pad_mode == 1;
    constexpr unsigned VEC = 4 * sizeof(T), INVALID = 0xC0000000u;       // bytes of a 4-channel vector; an offset no plane reaches (planes are < 2 GiB)
    // Buffer addressing throughout: the source plane and this workgroup's weight blocks behind descriptors (SGPRs), 32-bit byte offsets per lane, a padding
    // pixel = an out-of-range offset that loads zeros, the weights' K-step as the load's scalar offset.  The fp32 MFMA runs at the fp32 VECTOR rate, i.e. on
    // the pipe every addressing instruction needs: the first cut (64-bit pointer arithmetic, selects, the pixel addressing recomputed per K-step) spent
    // 100 VALU instructions per 16-MFMA K-step and sat at MFMA busy 50 % + VALU 42 % (profiles/r5/precise_engine_error.txt).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(a.srcA)) + (a.shareA ? (size_t)0 : (size_t)s * a.HA * a.WA * a.CA * sizeof(T)), 0,
        (unsigned)((size_t)a.HA * a.WA * a.CA * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(VB ? a.srcB : a.srcA)) + ((a.shareB || !VB) ? (size_t)0 : (size_t)s * a.Hin * a.Win * a.CB * sizeof(T)), 0,
        (unsigned)(VB ? (size_t)a.Hin * a.Win * a.CB * sizeof(T) : 0), 0x00020000);
    __amdgpu_buffer_rsrc_t rsW[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        rsW[nb] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(a.wpack)) + (size_t)(bg * NB + nb) * nsteps * 64 * VEC, 0,
                                                    (unsigned)((size_t)nsteps * 64 * VEC), 0x00020000);
    const unsigned wl = (unsigned)lane * VEC;
    int step = 0;                                                        // K-step counter over both sources (weights, flush)
    auto flush = [&](const int st) {
        if constexpr (TWO_LEVEL) {
            if ((st & (FLUSH - 1)) == FLUSH - 1 || st == nsteps - 1) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pg = 0; pg < PG; ++pg) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) carry[nb][pg][i] += (double)acc[nb][pg][i];
                        acc[nb][pg] = zero;
                    }
            }
        }
    };
    // one source: its K-vectors v = 4 st + g = tap * Vs + c4 (tap = ky * ks + kx), advanced incrementally per lane; the byte offset of the tap's pixel in
    // every pixel group is recomputed only when the lane moves on to the next tap
    auto segment = [&](const __amdgpu_buffer_rsrc_t rs, const int Vs, const int nst, const int up, const int pitch) {
        const int tap0 = g / Vs;
        int c4 = g - tap0 * Vs, ky = tap0 / ks, kx = tap0 - ky * ks;
        unsigned poff[PG];
        auto tap_pixels = [&]() {
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                int iy = oy[pg] + ky, ix = ox[pg] + kx;
                bool ok = pv[pg] && ky < ks;                             // past the last tap: zero operands (the packed weights are zero there too)
                if (reflect) {                                           // nn.ReflectionPad2d(1), model/CPN/decoder.py:23
                    iy = iy < 0 ? -iy : (iy >= a.Hin ? 2 * a.Hin - 2 - iy : iy);
                    ix = ix < 0 ? -ix : (ix >= a.Win ? 2 * a.Win - 2 - ix : ix);
                } else {
                    ok = ok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                }
                poff[pg] = ok ? (unsigned)((iy >> up) * pitch + (ix >> up)) * ((unsigned)Vs * VEC) : INVALID;
            }
        };
        tap_pixels();
        for (int st = 0; st < nst; ++st, ++step) {
            const unsigned cb = (unsigned)c4 * VEC, wso = (unsigned)step * (64u * VEC);
            v4 xv[PG], wv[NB];
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) xv[pg] = buf_load4(rs, poff[pg] + cb, 0u, T());
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wv[nb] = buf_load4(rsW[nb], wl, wso, T());
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pg = 0; pg < PG; ++pg) acc[nb][pg] = mfma16(wv[nb][j], xv[pg][j], acc[nb][pg]);
            flush(step);
            c4 += 4;
            if (c4 >= Vs) {
                do {
                    c4 -= Vs;
                    if (++kx == ks) { kx = 0; ++ky; }
                } while (c4 >= Vs);
                tap_pixels();
            }
        }
    };
    segment(rsA, VA, nstA, a.up, a.WA);
    if (VB) segment(rsB, VB, nstB, 0, a.Win);
    if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[nb][pg][i] = (T)carry[nb][pg][i];       // ONE rounding of the whole sum to fp32
    }
    int pix[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) pix[pg] = p0 + 16 * pg + m;
    pconv_epilogue<T, NB, PG>(a, acc, pix, g, bg, s, P);
}

// ---- fp32-grade products on the bf16 matrix cores ("x3") -------------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate and on the vector pipe (157 TFLOP/s, shared with every addressing instruction);
// v_mfma_f32_16x16x32_bf16 is sixteen times faster and has the matrix pipe to itself.  An fp32 number is EXACTLY the sum of three bf16
// numbers (a1 = bf16(a), a2 = bf16(a - a1), a3 = a - a1 - a2: both subtractions are exact in fp32 and the last residual has at most
// 8 significant bits), bf16 x bf16 products are exact in the instruction's fp32 accumulation, so
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1) + (a2 b3 + a3 b2) + a3 b3
// with the terms of relative size 1, 2^-8, 2^-16, 2^-24, 2^-32.  TERMS = 6 drops the last three (a relative 2^-24 per product, the size of
// the rounding of an fp32 product, signed and unbiased because the split rounds to nearest), TERMS = 8 only a3 b3.  The leading products go
// into one accumulator, the small ones into a second one, both are added into the fp64 carries every FLUSH steps (64 leading products):
// the same two-level sum as k_pconv<float>.  Weights are split on the host (pack_weights_x3), activations in the loader: 9 VALU
// instructions per pair of elements (v_cvt_pk_bf16_f32 x3, two shifts / masks x2, v_pk_add_f32 x2), shared by the NB row blocks.
// The barrier of the kernels that copy weights global -> LDS (LDS-DMA): a copy is ordered for other waves' reads only by the ISSUING wave's vmcnt wait followed by a
// barrier.  __syncthreads()'s fence emits that wait only where the compiler believes a copy is pending - it lost track of one issued under a wave-dependent
// condition (k_pconv_x3_chunk<1>: a bare s_barrier, stale weights, caught by the forced-form test) - so the wait is spelled out.
#define MPF_COPY_BARRIER()                                  \
    do {                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
        __syncthreads();                                    \
    } while (0)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ Vec4<float>::type mfma_bf16(const u32x4_t a, const u32x4_t b, const Vec4<float>::type c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// two fp32 values -> their three bf16 pieces, packed (low half = the first value)
struct Pieces { unsigned p[3]; };
__device__ __forceinline__ Pieces split3(const float a0, const float a1)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    Pieces r;
    const f2 a = {a0, a1};
    r.p[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_t));
    const f2 ra = a - f2{__builtin_bit_cast(float, r.p[0] << 16), __builtin_bit_cast(float, r.p[0] & 0xffff0000u)};      // exact (v_pk_add_f32)
    r.p[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(ra, bf16x2_t));
    const f2 rb = ra - f2{__builtin_bit_cast(float, r.p[1] << 16), __builtin_bit_cast(float, r.p[1] & 0xffff0000u)};     // exact, at most 8 significant bits left
    r.p[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(rb, bf16x2_t));
    return r;
}

template <int NB, int PG, int TERMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_pconv_x3(const MpfPConvArgs a)       // <= 256 registers: the MFMAs take their accumulators in VGPRs (no AGPR copies around the carries)
{
    typedef Vec4<float>::type v4;
    typedef Vec4<double>::type v4d;
    constexpr unsigned VEC = 16, INVALID = 0xC0000000u, WSTEP = 3 * 64 * 16;      // WSTEP: bytes of one step of one row block = three pieces x 64 lanes x 8 bf16
    constexpr int MAXTAP = 9;
    // the four waves of a workgroup (different pixels, the SAME NB row blocks) share the weights: every step's NB x 3 fragments go global -> LDS once per
    // workgroup (LDS-DMA, no registers), two buffers, one barrier per step.  Per-wave fragment loads made the kernel L1-bound: 10 KB of operands per 24 MFMAs
    // = 104 B / clock / CU against the ~57 the vector-memory path delivers (profiles/r5/precise_x3.txt)
    __shared__ __attribute__((aligned(16))) char wlds[2 * NB * WSTEP];
    // the byte offset of every tap's pixel, per lane and pixel group: computed once per source, looked up per step (the bounds / reflection logic of
    // k_pconv's cursor ran on every step - divergent, ~40 instructions - and the VALU is what this kernel has least of)
    __shared__ unsigned ptab[4][MAXTAP + 1][PG][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, g = lane >> 4;
    const int P = a.Hout * a.Wout;
    const int p0 = ((int)blockIdx.x * 4 + wave) * (16 * PG);             // a wave past the last pixel stays for the barriers and the weight copies
    const int bg = blockIdx.y, s = blockIdx.z;
    const int VA = a.CA >> 2, VB = a.CB >> 2, ks = a.ksize, ntap = ks * ks;
    // a step = 32 K elements = TWO of k_pconv's K-steps: lane (m, g) holds the 4 channels of K-vector 4 (2 t) + g, then those of K-vector 4 (2 t + 1) + g;
    // every source is padded to an even number of the old steps (zero weights; the cursor is past the last tap there and loads zeros)
    const int n2A = (((ntap * VA + 3) >> 2) + 1) >> 1, n2B = (((ntap * VB + 3) >> 2) + 1) >> 1, nsteps = n2A + n2B;
    int oy[PG], ox[PG];
    bool pv[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int p = p0 + 16 * pg + m;
        pv[pg] = p < P;
        const int pc = pv[pg] ? p : 0;
        oy[pg] = pc / a.Wout;
        ox[pg] = pc - oy[pg] * a.Wout;
        oy[pg] = oy[pg] * a.stride - a.pad;
        ox[pg] = ox[pg] * a.stride - a.pad;
    }
    const v4 zero = {0.f, 0.f, 0.f, 0.f};
    // accH: the leading products a1 b1 of TWO steps (64 products), then added into the fp64 carries - the two-level sum of k_pconv<float>; the first of the
    // two steps starts from the instruction's zero operand, so the accumulators are never cleared.  accL: everything else - 2^-8 of the leading sum and
    // below, so its own fp32 roundings are 2^-32 of the result - runs through the whole K loop and is added once.
    v4 accH[NB][PG], accL[NB][PG];
    v4d carry[NB][PG];
    const bool gated = a.epi == EP_GATED || a.epi == EP_GATED_PLANAR;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        v4 init = zero;
        if (gated) init = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(a.bias) + (bg * NB + nb) * 16 + 4 * g);
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            carry[nb][pg] = v4d{(double)init[0], (double)init[1], (double)init[2], (double)init[3]};
            accH[nb][pg] = zero;
            accL[nb][pg] = zero;
        }
    }
    const bool reflect = a.pad_mode == 1;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(a.srcA)) + (a.shareA ? (size_t)0 : (size_t)s * a.HA * a.WA * a.CA * 4), 0, (unsigned)((size_t)a.HA * a.WA * a.CA * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(VB ? a.srcB : a.srcA)) + ((a.shareB || !VB) ? (size_t)0 : (size_t)s * a.Hin * a.Win * a.CB * 4), 0,
        (unsigned)(VB ? (size_t)a.Hin * a.Win * a.CB * 4 : 0), 0x00020000);
    // weights: fragment f = nb * 3 + piece of step t lives at ((bg NB + nb) nsteps + t) WSTEP + piece KB; wave w copies fragments w, w + 4, ...
    const char *wg = reinterpret_cast<const char *>(a.wpack) + (size_t)bg * NB * nsteps * WSTEP + (unsigned)lane * 16u;
    int wreq = 0;                                                        // the next step whose weights have not been requested
    auto weights_upto = [&](const int t) {
        for (; wreq <= t && wreq < nsteps; ++wreq) {
#pragma unroll
            for (int j = 0; j < (NB * 3 + 3) / 4; ++j) {
                const int f = wave + 4 * j;                              // wave-uniform
                if ((NB * 3) % 4 == 0 || f < NB * 3) {
                    const int nb = f / 3, q = f - 3 * nb;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wg + ((size_t)(nb * nsteps + wreq) * WSTEP + (unsigned)q * 1024u)),
                                                     (__attribute__((address_space(3))) void *)(wlds + ((wreq & 1) * NB * 3 + f) * 1024), 16, 0, 0);
                }
            }
        }
    };
    struct Pixels { v4 x0[PG], x1[PG]; };
    // FIRST: the first of the two steps between flushes (leading accumulators start from zero)
    auto compute = [&](const Pixels &o, const int step, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        u32x4_t xp[PG][3];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
#if MPF_X3_ABLATE == 2                                                   // timing ablation ONLY (tools/build_ablate_x3.sh): no split
#pragma unroll
            for (int q = 0; q < 3; ++q) xp[pg][q] = u32x4_t{__builtin_bit_cast(unsigned, o.x0[pg][q]), __builtin_bit_cast(unsigned, o.x0[pg][3]), __builtin_bit_cast(unsigned, o.x1[pg][q]), __builtin_bit_cast(unsigned, o.x1[pg][3])};
#else
            const Pieces q0 = split3(o.x0[pg][0], o.x0[pg][1]), q1 = split3(o.x0[pg][2], o.x0[pg][3]), q2 = split3(o.x1[pg][0], o.x1[pg][1]), q3 = split3(o.x1[pg][2], o.x1[pg][3]);
#pragma unroll
            for (int q = 0; q < 3; ++q) xp[pg][q] = u32x4_t{q0.p[q], q1.p[q], q2.p[q], q3.p[q]};
#endif
        }
        const char *wb = wlds + (step & 1) * (NB * WSTEP) + lane * 16;
        // by weight piece (its NB fragments are read from LDS once); the terms of the FIRST activation piece first - it is ready after one conversion, the
        // matrix pipe starts while the vector pipe still splits; NB x PG independent accumulators between dependent issues
#pragma unroll
        for (int qa = 2; qa >= 0; --qa) {
            u32x4_t w[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) w[nb] = *reinterpret_cast<const u32x4_t *>(wb + (nb * 3 + qa) * 1024);
#pragma unroll
            for (int qb = 0; qb < 3; ++qb) {
                if (qa + qb > (TERMS == 8 ? 3 : 2)) continue;            // six terms: pieces (a, b) with a + b <= 2; eight: all but (2, 2)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pg = 0; pg < PG; ++pg) {
#if MPF_X3_ABLATE == 1                                                   // timing ablation ONLY: no MFMA (one VALU op keeps the operands alive)
                        if (qa + qb == 0) accH[nb][pg][0] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, FIRST ? 0.f : accH[nb][pg][0]) ^ w[nb][0] ^ xp[pg][qb][0]);
                        else if (qa + qb == 2 && qa == 1) accL[nb][pg][0] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, accL[nb][pg][0]) ^ w[nb][1] ^ xp[pg][qb][1] ^ xp[pg][2][2] ^ w[nb][3]);
#else
                        if (qa + qb == 0) accH[nb][pg] = mfma_bf16(w[nb], xp[pg][qb], FIRST ? zero : accH[nb][pg]);
                        else accL[nb][pg] = mfma_bf16(w[nb], xp[pg][qb], accL[nb][pg]);
#endif
                    }
            }
        }
    };
    auto flush = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg)
#pragma unroll
                for (int i = 0; i < 4; ++i) carry[nb][pg][i] += (double)accH[nb][pg][i];
    };
    int step = 0;
    // one source: its K-vectors v = tap * Vs + c4, four per old step, two old steps per step.  The pixels and the weights of step t + 1 are requested before
    // step t is computed (two register sets / two LDS buffers, the loop unrolled by two); the barrier at the top of a step (hipcc drains vmcnt before it) makes
    // step t's weights visible to every wave and says that every wave is done reading the buffer step t + 1's weights go to.  The pixel request past a
    // source's last step reads zeros and is dropped.
    auto segment = [&](const __amdgpu_buffer_rsrc_t rs, const int Vs, const int n2, const int up, const int pitch) {
        // tap table of this source (a wave reads only what it wrote: no barrier; the previous source's last lookups are done - their loads were issued)
        for (int t = 0; t <= ntap; ++t) {
            const int ky = t / ks, kx = t - ky * ks;
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                int iy = oy[pg] + ky, ix = ox[pg] + kx;
                bool ok = pv[pg] && t < ntap;                            // past the last tap: zero operands (the packed weights are zero there too)
                if (reflect) {                                           // nn.ReflectionPad2d(1), model/CPN/decoder.py:23
                    iy = iy < 0 ? -iy : (iy >= a.Hin ? 2 * a.Hin - 2 - iy : iy);
                    ix = ix < 0 ? -ix : (ix >= a.Win ? 2 * a.Win - 2 - ix : ix);
                } else {
                    ok = ok && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                }
                ptab[wave][t][pg][lane] = ok ? (unsigned)((iy >> up) * pitch + (ix >> up)) * ((unsigned)Vs * VEC) : INVALID;
            }
        }
        int tp = g / Vs, c4 = g - tp * Vs;
        auto request = [&](v4 (&x)[PG]) {
            const int tc = tp < ntap ? tp : ntap;
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) x[pg] = buf_load4(rs, ptab[wave][tc][pg][lane] + (unsigned)c4 * VEC, 0u, float());
            c4 += 4;
            if (Vs >= 4) {                                               // uniform
                const bool wrap = c4 >= Vs;
                c4 -= wrap ? Vs : 0;
                tp += wrap ? 1 : 0;
            } else {
                while (c4 >= Vs) { c4 -= Vs; ++tp; }
            }
        };
        Pixels A, B;
        request(A.x0);
        request(A.x1);
        weights_upto(step);
        for (int st = 0; st < n2; st += 2) {
#if MPF_X3_ABLATE != 3                                                   // timing ablation ONLY: no barrier
            MPF_COPY_BARRIER();
#endif
            request(B.x0);
            request(B.x1);
            weights_upto(step + 1);
            compute(A, step++, std::true_type());
            if (st + 1 < n2) {
#if MPF_X3_ABLATE != 3
                MPF_COPY_BARRIER();
#endif
                request(A.x0);
                request(A.x1);
                weights_upto(step + 1);
                compute(B, step++, std::false_type());
            }
            flush();
        }
    };
    segment(rsA, VA, n2A, a.up, a.WA);
    if (VB) segment(rsB, VB, n2B, 0, a.Win);
    if (p0 >= P) return;
    v4 acc[NB][PG];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int pg = 0; pg < PG; ++pg)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nb][pg][i] = (float)(carry[nb][pg][i] + (double)accL[nb][pg][i]);        // ONE rounding of the whole sum to fp32
    int pix[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) pix[pg] = p0 + 16 * pg + m;
    pconv_epilogue<float, NB, PG>(a, acc, pix, g, bg, s, P);
}

// The few-channel layers at full resolution (the UNet's first / last levels, the decoder's last level: a third of the forward) have a dozen K steps per wave
// and spend them loading, bounds-checking and splitting every input element nine times - once per tap.  k_pconv_x3_tile: 3 x 3, stride 1, padding 1, at most
// 56 input channels and 3 row blocks.  A workgroup owns an 8 x 16 pixel tile of one plane: its 10 x 18 halo of BOTH sources (concatenated, the first one
// optionally x2 nearest up-sampled, padding resolved) is loaded and split ONCE into LDS as [pixel][8-channel vector][piece][8 bf16]; the K loop - K-vector
// 4 t + g = (tap, 8-channel vector) - reads its three activation pieces with one ds_read_b128 each and has no bounds logic left.  Weights per
// wave from global memory (registers, two steps ahead) in the K order of this kernel (pack_weights_x3_tile).
constexpr int TILE_H = 8, TILE_W = 16, HALO_W = TILE_W + 2, HALO_PIX = (TILE_H + 2) * HALO_W;
__host__ __device__ constexpr int tile_pix_bytes(int V8) { return V8 * 48 + (V8 % 2 == 0 ? 16 : 0); }      // 4 x odd dwords: 16 consecutive pixels' 16-byte reads cover the 64 banks once

// PH (MpfPConvArgs.up == 2, round 6): the x2-nearest, reflection-padded layer WITHOUT a second source (upconv(0,1), model/CPN/decoder.py:19-20,160-162) phase-decomposed as in
// the fast engine (mpf_conv.hip: k_conv3x3_up): a workgroup owns an 8 x 16 tile of LOW-RESOLUTION cells and ONE output phase (py, px); its 10 x 18 halo is the
// low-resolution map CLAMPED at the border (= the reflection of the upsampled map), the K loop runs the four taps (py + ty, px + tx) of the halo with that phase's
// weights - sums of the nine, taken in float64 on the host and split into the same three bf16 pieces - and the outputs go to (2 y + py, 2 x + px).  4 taps instead of 9
// and a quarter of the staging per output; the products and the two-level sum are those of the gather form.
template <int NB, int TERMS, bool PH = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_pconv_x3_tile(const MpfPConvArgs a)
{
    typedef Vec4<float>::type v4;
    typedef Vec4<double>::type v4d;
    constexpr int PG = 2, NTAP = PH ? 4 : 9;
    constexpr unsigned WSTEP = 3 * 64 * 16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *const tile = lds;                                              // [HALO_PIX][PIXB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, g = lane >> 4;
    const int P = a.Hout * a.Wout;
    const int tx0 = blockIdx.x * TILE_W, ty0 = blockIdx.y * TILE_H;        // PH: in low-resolution cells
    const int zz = PH ? (int)(blockIdx.z >> 2) : (int)blockIdx.z, ph = PH ? (int)(blockIdx.z & 3u) : 0, py = ph >> 1, px = ph & 1;
    const int nbg = a.nblk / NB, s = zz / nbg, bg = zz - s * nbg;
    const int VA = a.CA >> 2, VT = (a.CA + a.CB) >> 2, V8 = (VT + 1) >> 1, PIXB = tile_pix_bytes(V8);
    const int nsteps = (NTAP * V8 + 3) >> 2;
    // ---- weights: per wave, global / L2 -> registers, requested two steps ahead (three register sets, the K loop unrolled by three): with a dozen steps per
    // tile a workgroup-shared LDS copy costs one barrier + one exposed L2 latency per step (measured: 2.5 x the MFMA time of the layer)
    __amdgpu_buffer_rsrc_t rsW[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        rsW[nb] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(a.wpack)) + (size_t)((ph * a.nblk) + bg * NB + nb) * nsteps * WSTEP, 0,
                                                    (unsigned)((size_t)nsteps * WSTEP), 0x00020000);          // PH: [phase][row block][step]
    struct Weights { u32x4_t w[NB][3]; };
    auto weights = [&](Weights &o, const int t) {                        // past the last step: out of the descriptor's range, zeros
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 3; ++q) o.w[nb][q] = __builtin_amdgcn_raw_buffer_load_b128(rsW[nb], (unsigned)lane * 16u + (unsigned)q * 1024u, (unsigned)t * WSTEP, 0);
    };
    Weights W0, W1, W2;
    weights(W0, 0);
    weights(W1, 1);
    // ---- the halo tile, one source after the other (a uniform buffer descriptor per pass; a padding pixel / a zero vector = an out-of-range offset);
    // when the vector count is odd the pass of the last source also writes the zero vector that completes the last 8-channel vector
    {
        const bool reflect = a.pad_mode == 1;
        auto stage = [&](const __amdgpu_buffer_rsrc_t rs, const int Vsrc, const int Viter, const int vbase, const int up, const int pitch) {
            const int items = HALO_PIX * Viter;
            const unsigned inv = (65536u + (unsigned)Viter - 1u) / (unsigned)Viter;      // i / Viter for i < 4096, Viter <= 15 (i * Viter < 2^16)
            constexpr int BATCH = 4;                                     // loads in flight per thread: the staging is latency, not bandwidth
            for (int i0 = tid; i0 < items; i0 += 256 * BATCH) {
                v4 x[BATCH];
                int dst[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int i = i0 + 256 * k;
                    const int px = (int)(((unsigned)i * inv) >> 16), v = i - px * Viter, hy = px / HALO_W, hx = px - hy * HALO_W;
                    int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
                    bool ok;
                    if constexpr (PH) {                                  // the low-resolution map clamped at its border = reflection padding of the upsampled one
                        iy = iy < 0 ? 0 : (iy > a.HA - 1 ? a.HA - 1 : iy);
                        ix = ix < 0 ? 0 : (ix > a.WA - 1 ? a.WA - 1 : ix);
                        ok = i < items && v < Vsrc;
                    } else {
                        if (reflect) {                                   // nn.ReflectionPad2d(1); rows / columns past the image's mirror line belong to no output pixel
                            iy = iy < 0 ? -iy : (iy >= a.Hin ? 2 * a.Hin - 2 - iy : iy);
                            ix = ix < 0 ? -ix : (ix >= a.Win ? 2 * a.Win - 2 - ix : ix);
                        }
                        ok = i < items && v < Vsrc && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                    }
                    x[k] = buf_load4(rs, ok ? (unsigned)(((iy >> up) * pitch + (ix >> up)) * Vsrc + v) * 16u : 0xC0000000u, 0u, float());
                    dst[k] = i < items ? px * PIXB + ((vbase + v) >> 1) * 48 + ((vbase + v) & 1) * 8 : -1;
                }
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    if (dst[k] < 0) continue;
                    const Pieces q0 = split3(x[k][0], x[k][1]), q1 = split3(x[k][2], x[k][3]);
#pragma unroll
                    for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2 *>(tile + dst[k] + q * 16) = make_uint2(q0.p[q], q1.p[q]);
                }
            }
        };
        const int VB = VT - VA, odd = VT & 1;
        stage(__builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(a.srcA)) + (a.shareA ? (size_t)0 : (size_t)s * a.HA * a.WA * a.CA * 4), 0,
                                                (unsigned)((size_t)a.HA * a.WA * a.CA * 4), 0x00020000), VA, VA + (VB ? 0 : odd), 0, PH ? 0 : a.up, a.WA);
        if (VB)
            stage(__builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(a.srcB)) + (a.shareB ? (size_t)0 : (size_t)s * a.Hin * a.Win * a.CB * 4), 0,
                                                    (unsigned)((size_t)a.Hin * a.Win * a.CB * 4), 0x00020000), VB, VB + odd, VA, 0, a.Win);
    }
    const v4 zero = {0.f, 0.f, 0.f, 0.f};
    v4 accH[NB][PG], accL[NB][PG];
    v4d carry[NB][PG];
    const bool gated = a.epi == EP_GATED || a.epi == EP_GATED_PLANAR;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        v4 init = zero;
        if (gated) init = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(a.bias) + (bg * NB + nb) * 16 + 4 * g);
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            carry[nb][pg] = v4d{(double)init[0], (double)init[1], (double)init[2], (double)init[3]};
            accH[nb][pg] = zero;
            accL[nb][pg] = zero;
        }
    }
    // lane (m, g) of wave w: output pixels (ty0 + 2 w + pg, tx0 + m); K-vector 4 t + g = tap * V8 + c8
    const int tap0 = g / V8;
    int c8 = g - tap0 * V8, tap = tap0;
    const char *xb = tile + ((2 * wave) * HALO_W + m) * PIXB;
    __syncthreads();                                                     // the tile is in LDS; no barrier from here on
    int parity = 0;                                                      // accH: two steps (64 leading products) between flushes, the first from the zero operand
    auto compute = [&](const Weights &o) {
        const int tc = tap < NTAP ? tap : NTAP - 1;                                  // past the last tap the weights are zero: any finite operand will do
        const int ky = PH ? py + (tc >> 1) : tc / 3, kx = PH ? px + (tc & 1) : tc - 3 * (tc / 3);
        const char *xl = xb + (ky * HALO_W + kx) * PIXB + c8 * 48;
        u32x4_t xp[PG][3];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg)
#pragma unroll
            for (int q = 0; q < 3; ++q) xp[pg][q] = *reinterpret_cast<const u32x4_t *>(xl + pg * (HALO_W * PIXB) + q * 16);
#pragma unroll
        for (int qa = 2; qa >= 0; --qa)
#pragma unroll
            for (int qb = 2; qb >= 0; --qb) {
                if (qa + qb > (TERMS == 8 ? 3 : 2)) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pg = 0; pg < PG; ++pg) {
                        if (qa + qb == 0) accH[nb][pg] = mfma_bf16(o.w[nb][qa], xp[pg][qb], accH[nb][pg]);
                        else accL[nb][pg] = mfma_bf16(o.w[nb][qa], xp[pg][qb], accL[nb][pg]);
                    }
            }
        if (parity) {                                                    // uniform
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int pg = 0; pg < PG; ++pg) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) carry[nb][pg][i] += (double)accH[nb][pg][i];
                    accH[nb][pg] = zero;
                }
        }
        parity ^= 1;
        c8 += 4;
        while (c8 >= V8) { c8 -= V8; ++tap; }
    };
    for (int step = 0; step < nsteps; step += 3) {
        weights(W2, step + 2);
        compute(W0);
        if (step + 1 < nsteps) {
            weights(W0, step + 3);
            compute(W1);
        }
        if (step + 2 < nsteps) {
            weights(W1, step + 4);
            compute(W2);
        }
    }
    v4 acc[NB][PG];
    int pix[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int cy = ty0 + 2 * wave + pg, cx = tx0 + m;                    // PH: the low-resolution cell; its output pixel of this phase
        const int oy = PH ? 2 * cy + py : cy, ox = PH ? 2 * cx + px : cx;
        pix[pg] = (oy < a.Hout && ox < a.Wout) ? oy * a.Wout + ox : P;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nb][pg][i] = (float)(carry[nb][pg][i] + ((double)accH[nb][pg][i] + (double)accL[nb][pg][i]));
    }
    pconv_epilogue<float, NB, PG>(a, acc, pix, g, bg, s, P);
}

// The many-channel 3 x 3 / stride 1 layers (the UNet's middle levels, the decoder above the last level): k_pconv_x3 loads, bounds-checks and splits every input
// element once per TAP and per row group - the split is half of its vector instructions, and on this kernel matrix time and vector time ADD UP
// (profiles/r5/precise_x3.txt).  k_pconv_x3_chunk is k_pconv_x3_tile with a loop over 32-channel chunks of the concatenated sources: a workgroup owns an 8 x 16 pixel
// tile and NB row blocks; per chunk the 10 x 18 halo is loaded and split ONCE into LDS ([pixel][4 x 8-channel vector][piece][8 bf16]), then nine steps - one per tap,
// lane group g = 8-channel vector g of the chunk, so the tap offset is a scalar and there is no per-lane cursor at all - read activations (3 ds_read_b128 per pixel
// group) and weights (LDS-DMA, two buffers, as k_pconv_x3) from LDS.  Channels are zero-padded to a multiple of 32 (pack_weights_x3_chunk: step = chunk * 9 + tap).
constexpr int CHUNK_PIXB = 4 * 48 + 16;                                  // 52 dwords = 4 x odd

template <int NB, int TERMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_pconv_x3_chunk(const MpfPConvArgs a)
{
    typedef Vec4<float>::type v4;
    typedef Vec4<double>::type v4d;
    constexpr int PG = 2;
    constexpr unsigned WSTEP = 3 * 64 * 16;
    __shared__ __attribute__((aligned(16))) char wlds[2 * NB * WSTEP];
    __shared__ __attribute__((aligned(16))) char tile[HALO_PIX * CHUNK_PIXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, g = lane >> 4;
    const int P = a.Hout * a.Wout;
    const int tx0 = blockIdx.x * TILE_W, ty0 = blockIdx.y * TILE_H;
    const int nbg = a.nblk / NB, s = blockIdx.z / nbg, bg = blockIdx.z - s * nbg;
    const int VA = a.CA >> 2, VT = (a.CA + a.CB) >> 2, nchunk = (VT + 7) >> 3, nsteps = nchunk * 9;
    const char *wg = reinterpret_cast<const char *>(a.wpack) + (size_t)bg * NB * nsteps * WSTEP + (unsigned)lane * 16u;
    auto weights = [&](const int t) {                                    // fragment f = nb * 3 + piece of step t; wave w copies fragments w, w + 4, ...
        if (t >= nsteps) return;
#pragma unroll
        for (int j = 0; j < (NB * 3 + 3) / 4; ++j) {
            const int f = wave + 4 * j;
            if ((NB * 3) % 4 == 0 || f < NB * 3) {
                const int nb = f / 3, q = f - 3 * nb;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wg + ((size_t)(nb * nsteps + t) * WSTEP + (unsigned)q * 1024u)),
                                                 (__attribute__((address_space(3))) void *)(wlds + ((t & 1) * NB * 3 + f) * 1024), 16, 0, 0);
            }
        }
    };
    const v4 zero = {0.f, 0.f, 0.f, 0.f};
    v4 accH[NB][PG], accL[NB][PG];
    v4d carry[NB][PG];
    const bool gated = a.epi == EP_GATED || a.epi == EP_GATED_PLANAR;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        v4 init = zero;
        if (gated) init = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(a.bias) + (bg * NB + nb) * 16 + 4 * g);
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            carry[nb][pg] = v4d{(double)init[0], (double)init[1], (double)init[2], (double)init[3]};
            accH[nb][pg] = zero;
            accL[nb][pg] = zero;
        }
    }
    // ---- staging of one chunk: item = (halo pixel, 4-channel vector j of the chunk); the pixel part of the addressing is the same for every chunk
    const bool reflect = a.pad_mode == 1;
    const float *srcA = reinterpret_cast<const float *>(a.srcA) + (a.shareA ? (size_t)0 : (size_t)s * a.HA * a.WA * a.CA);
    const float *srcB = reinterpret_cast<const float *>(a.CB ? a.srcB : a.srcA) + ((a.shareB || !a.CB) ? (size_t)0 : (size_t)s * a.Hin * a.Win * a.CB);
    constexpr int NIT = (HALO_PIX * 8 + 255) / 256;                      // 6 items per thread and chunk
    auto stage = [&](const int c) {
        v4 x[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            // the pixel part of the addressing is recomputed per chunk: twelve registers held across the K loop cost more (spills at four row blocks)
            const int i = tid + 256 * k, px = i >> 3, hy = px / HALO_W, hx = px - hy * HALO_W, v = 8 * c + (i & 7);
            int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            if (reflect) {                                               // nn.ReflectionPad2d(1)
                iy = iy < 0 ? -iy : (iy >= a.Hin ? 2 * a.Hin - 2 - iy : iy);
                ix = ix < 0 ? -ix : (ix >= a.Win ? 2 * a.Win - 2 - ix : ix);
            }
            const bool ok = px < HALO_PIX && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win && v < VT;
            x[k] = zero;
            if (ok) x[k] = v < VA ? reinterpret_cast<const v4 *>(srcA)[(unsigned)((iy >> a.up) * a.WA + (ix >> a.up)) * (unsigned)VA + (unsigned)v]
                                  : reinterpret_cast<const v4 *>(srcB)[(unsigned)(iy * a.Win + ix) * (unsigned)(VT - VA) + (unsigned)(v - VA)];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + 256 * k, px = i >> 3, j = i & 7;
            if (NIT * 256 != HALO_PIX * 8 && px >= HALO_PIX) continue;
            const Pieces q0 = split3(x[k][0], x[k][1]), q1 = split3(x[k][2], x[k][3]);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2 *>(tile + px * CHUNK_PIXB + (j >> 1) * 48 + (j & 1) * 8 + q * 16) = make_uint2(q0.p[q], q1.p[q]);
        }
    };
    // lane (m, g) of wave w: output pixels (ty0 + 2 w + pg, tx0 + m); step t = chunk * 9 + tap, K-vector g of the step = 8-channel vector g of the chunk
    const char *xb = tile + ((2 * wave) * HALO_W + m) * CHUNK_PIXB + g * 48;
    auto compute = [&](const int t, const int tap, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const char *xl = xb + (ky * HALO_W + kx) * CHUNK_PIXB;
        u32x4_t xp[PG][3];
#pragma unroll
        for (int pg = 0; pg < PG; ++pg)
#pragma unroll
            for (int q = 0; q < 3; ++q) xp[pg][q] = *reinterpret_cast<const u32x4_t *>(xl + pg * (HALO_W * CHUNK_PIXB) + q * 16);
        // weight fragments by inline assembly: after a global_load_lds the compiler puts s_waitcnt vmcnt(0) in front of every LDS read it can see that may alias the
        // copy's destination - that would wait for the copy of step t + 1, just issued, before step t starts.  The barriers order these reads against the copies.
        const unsigned wb = (unsigned)(uintptr_t)(wlds + (t & 1) * (NB * WSTEP) + lane * 16);
#pragma unroll
        for (int qa = 2; qa >= 0; --qa) {
            u32x4_t w[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[nb]) : "v"(wb), "n"((nb * 3 + qa) * 1024));
            if constexpr (NB == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
            else if constexpr (NB == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]));
            else if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]));
#pragma unroll
            for (int qb = 2; qb >= 0; --qb) {
                if (qa + qb > (TERMS == 8 ? 3 : 2)) continue;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pg = 0; pg < PG; ++pg) {
                        if (qa + qb == 0) accH[nb][pg] = mfma_bf16(w[nb], xp[pg][qb], FIRST ? zero : accH[nb][pg]);
                        else accL[nb][pg] = mfma_bf16(w[nb], xp[pg][qb], accL[nb][pg]);
                    }
            }
        }
    };
    auto flush = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg)
#pragma unroll
                for (int i = 0; i < 4; ++i) carry[nb][pg][i] += (double)accH[nb][pg][i];
    };
    weights(0);
    for (int c = 0; c < nchunk; ++c) {
        if (c) MPF_COPY_BARRIER();                                       // every wave is done with the previous chunk's tile
        stage(c);
        // the barrier at the top of a step: this step's weights (and, at tap 0, the tile) are in LDS; the other weight buffer is free.
        // accH: taps (0,1) (2,3) (4,5) (6,7) in pairs - 64 leading products per flush, the first of a pair from the zero operand - tap 8 alone
        int t = c * 9;
#pragma unroll 1
        for (int tp = 0; tp < 4; ++tp) {
            MPF_COPY_BARRIER();
            weights(t + 1);
            compute(t, 2 * tp, std::true_type());
            ++t;
            MPF_COPY_BARRIER();
            weights(t + 1);
            compute(t, 2 * tp + 1, std::false_type());
            ++t;
            flush();
        }
        MPF_COPY_BARRIER();
        weights(t + 1);
        compute(t, 8, std::true_type());
        flush();
    }
    v4 acc[NB][PG];
    int pix[PG];
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int oy = ty0 + 2 * wave + pg, ox = tx0 + m;
        pix[pg] = (oy < a.Hout && ox < a.Wout) ? oy * a.Wout + ox : P;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[nb][pg][i] = (float)(carry[nb][pg][i] + (double)accL[nb][pg][i]);
    }
    pconv_epilogue<float, NB, PG>(a, acc, pix, g, bg, s, P);
}

// ---- the tensors the reference builds with expand / cat / Upsample / adaptive_avg_pool2d, materialised ---------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fmn_input(const float *__restrict__ image, const float *__restrict__ disp, const float *__restrict__ plane_vals, int S, int N,
                                                   T *__restrict__ out)
{
    // model/CPN/unet.py:44-50: cat(image, disparity, plane disparity) per plane -> [S,H,W,8] (channels 5..7 zero)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)S * N) return;
    const int s = (int)(i / N), n = (int)(i - (size_t)s * N);
    T *o = out + i * 8;
    o[0] = (T)image[n]; o[1] = (T)image[N + n]; o[2] = (T)image[2 * N + n]; o[3] = (T)disp[n]; o[4] = (T)plane_vals[s];
    o[5] = (T)0; o[6] = (T)0; o[7] = (T)0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_encoder_input(const float *__restrict__ image, const float *__restrict__ disp, int N, T *__restrict__ out)
{
    // model/CPN/encoder.py:89-93: cat((image - mean) / std, disparity); the fp32 ImageNet constants of :84-85
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    T *o = out + (size_t)n * 4;
    o[0] = ((T)image[n] - (T)0.485f) / (T)0.229f;
    o[1] = ((T)image[N + n] - (T)0.456f) / (T)0.224f;
    o[2] = ((T)image[2 * N + n] - (T)0.406f) / (T)0.225f;
    o[3] = (T)disp[n];
}

template <typename T>
__global__ __launch_bounds__(256) void k_bilinear2x(const T *__restrict__ src, int S, int h, int w, int V, T *__restrict__ dst)
{
    // nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) (model/CPN/unet.py:42) on NHWC: src index = dst index * (in-1)/(out-1),
    // ATen's formula  h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
    typedef typename Vec4<T>::type v4;
    const int H = 2 * h, W = 2 * w;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)S * H * W * V) return;
    const int c4 = (int)(i % V);
    size_t r = i / V;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H), s = (int)(r / H);
    const T sy = H > 1 ? (T)(h - 1) / (T)(H - 1) : (T)0, sx = W > 1 ? (T)(w - 1) / (T)(W - 1) : (T)0;
    const T fy = sy * (T)y, fx = sx * (T)x;
    const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const T ly = fy - (T)y0, lx = fx - (T)x0, hy = (T)1 - ly, hx = (T)1 - lx;
    const v4 *p = reinterpret_cast<const v4 *>(src) + (size_t)s * h * w * V + c4;
    const v4 v00 = p[(size_t)(y0 * w + x0) * V], v01 = p[(size_t)(y0 * w + x1) * V], v10 = p[(size_t)(y1 * w + x0) * V], v11 = p[(size_t)(y1 * w + x1) * V];
    v4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = hy * (hx * v00[j] + lx * v01[j]) + ly * (hx * v10[j] + lx * v11[j]);
    reinterpret_cast<v4 *>(dst)[i] = o;
}

template <typename T>
__global__ __launch_bounds__(256) void k_per_plane(const T *__restrict__ feat, const T *__restrict__ cm, const T *__restrict__ fm, int S, int N, int VC, T *__restrict__ out)
{
    // model/CPN/decoder.py:140-150: a shared feature map expanded per plane: cat(feat * context_mask, context_mask, feature_mask) -> [S,h,w,C+4]
    typedef typename Vec4<T>::type v4;
    const int VO = VC + 1;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)S * N * VO) return;
    const int c4 = (int)(i % VO);
    const size_t sp = i / VO;
    const int n = (int)(sp % N);
    const T c = cm[sp];
    v4 o;
    if (c4 < VC) {
        const v4 f = reinterpret_cast<const v4 *>(feat)[(size_t)n * VC + c4];
        o = v4{f[0] * c, f[1] * c, f[2] * c, f[3] * c};
    } else {
        o = v4{c, fm[sp], (T)0, (T)0};
    }
    reinterpret_cast<v4 *>(out)[i] = o;
}

template <typename T>
__global__ __launch_bounds__(256) void k_softmax_planes(const T *__restrict__ logits, int S, int N, T *__restrict__ fmask, T *__restrict__ cum, T *__restrict__ ctx)
{
    // model/CPN/unet.py:68-69 softmax over the planes; model/CPN/decoder.py:127-129 cumulative mask and context mask 1 - cat(0, cum[:-1])
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    T mx = logits[n];
    for (int s = 1; s < S; ++s) mx = fmax(mx, logits[(size_t)s * N + n]);
    T sum = (T)0;
    for (int s = 0; s < S; ++s) sum += exp_t(logits[(size_t)s * N + n] - mx);
    T run = (T)0;
    for (int s = 0; s < S; ++s) {
        const T p = exp_t(logits[(size_t)s * N + n] - mx) / sum;
        ctx[(size_t)s * N + n] = (T)1 - run;
        run += p;
        cum[(size_t)s * N + n] = run;
        fmask[(size_t)s * N + n] = p;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_avgpool(const T *__restrict__ src, int S, int H, int W, int k, T *__restrict__ dst)
{
    // F.adaptive_avg_pool2d to (H/k, W/k) for H, W divisible by k: the mean of each k x k block (model/CPN/decoder.py:143-146)
    const int h = H / k, w = W / k;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)S * h * w) return;
    const int x = (int)(i % w);
    const size_t r = i / w;
    const int y = (int)(r % h), s = (int)(r / h);
    const T *p = src + ((size_t)s * H + (size_t)y * k) * W + (size_t)x * k;
    T sum = (T)0;
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) sum += p[(size_t)dy * W + dx];
    dst[i] = sum / (T)(k * k);
}

template <typename T>
__global__ __launch_bounds__(256) void k_maxpool3x3s2(const T *__restrict__ src, int Hin, int Win, int V, int Hout, int Wout, T *__restrict__ out)
{
    // nn.MaxPool2d(3, 2, 1) on NHWC (model/CPN/encoder.py:97 via the ResNet stem; model/CPN/decoder.py:81): padding never wins the max
    typedef typename Vec4<T>::type v4;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= Hout * Wout * V) return;
    const int c4 = n % V, p = n / V, oy = p / Wout, ox = p - oy * Wout;
    v4 best = {(T)-INFINITY, (T)-INFINITY, (T)-INFINITY, (T)-INFINITY};
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
        if (iy < 0 || iy >= Hin) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if (ix < 0 || ix >= Win) continue;
            const v4 t = reinterpret_cast<const v4 *>(src)[(size_t)(iy * Win + ix) * V + c4];
#pragma unroll
            for (int j = 0; j < 4; ++j) best[j] = best[j] > t[j] ? best[j] : t[j];
        }
    }
    reinterpret_cast<v4 *>(out)[n] = best;
}

template <typename T>
int launch_pconv(const MpfPConvArgs &a, hipStream_t st)
{
    const int P = a.Hout * a.Wout;
    if (a.nblk % 2 == 0) {
        hipLaunchKernelGGL((k_pconv<T, 2, 2>), dim3((P + 127) / 128, a.nblk / 2, a.S), dim3(256), 0, st, a);
    } else if (a.nblk % 3 == 0) {                                        // 24 gated channels = 48 rows: one pass over the input instead of three
        hipLaunchKernelGGL((k_pconv<T, 3, 2>), dim3((P + 127) / 128, a.nblk / 3, a.S), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_pconv<T, 1, 4>), dim3((P + 255) / 256, a.nblk, a.S), dim3(256), 0, st, a);
    }
    return mpf_launch_status("k_pconv");
}

template <int TERMS>
int launch_pconv_x3(const MpfPConvArgs &a, hipStream_t st)
{
    const int P = a.Hout * a.Wout;
    if (a.nblk % 4 == 0) {
        hipLaunchKernelGGL((k_pconv_x3<4, 2, TERMS>), dim3((P + 127) / 128, a.nblk / 4, a.S), dim3(256), 0, st, a);
    } else if (a.nblk % 3 == 0) {
        hipLaunchKernelGGL((k_pconv_x3<3, 2, TERMS>), dim3((P + 127) / 128, a.nblk / 3, a.S), dim3(256), 0, st, a);
    } else if (a.nblk % 2 == 0) {
        hipLaunchKernelGGL((k_pconv_x3<2, 2, TERMS>), dim3((P + 127) / 128, a.nblk / 2, a.S), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_pconv_x3<1, 4, TERMS>), dim3((P + 255) / 256, a.nblk, a.S), dim3(256), 0, st, a);
    }
    return mpf_launch_status("k_pconv_x3");
}

template <int NB>
int launch_pconv_x3_tile_nb(const MpfPConvArgs &a, hipStream_t st)
{
    const int V8 = ((a.CA + a.CB) / 4 + 1) / 2;
    const size_t lds = (size_t)HALO_PIX * tile_pix_bytes(V8);
    static bool attr_set = false;
    if (!attr_set) {
        MPF_HIP(hipFuncSetAttribute((const void *)k_pconv_x3_tile<NB, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    if (a.up == 2) {                                                     // phase-decomposed: tiles of low-resolution cells x four phases
        static bool attr_ph = false;
        if (!attr_ph) {
            MPF_HIP(hipFuncSetAttribute((const void *)k_pconv_x3_tile<NB, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            attr_ph = true;
        }
        hipLaunchKernelGGL((k_pconv_x3_tile<NB, 6, true>), dim3((a.WA + TILE_W - 1) / TILE_W, (a.HA + TILE_H - 1) / TILE_H, a.S * (a.nblk / NB) * 4), dim3(256), lds, st, a);
        return mpf_launch_status("k_pconv_x3_tile (phases)");
    }
    hipLaunchKernelGGL((k_pconv_x3_tile<NB, 6>), dim3((a.Wout + TILE_W - 1) / TILE_W, (a.Hout + TILE_H - 1) / TILE_H, a.S * (a.nblk / NB)), dim3(256), lds, st, a);
    return mpf_launch_status("k_pconv_x3_tile");
}

template <int NB>
int launch_pconv_x3_chunk_nb(const MpfPConvArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL((k_pconv_x3_chunk<NB, 6>), dim3((a.Wout + TILE_W - 1) / TILE_W, (a.Hout + TILE_H - 1) / TILE_H, a.S * (a.nblk / NB)), dim3(256), 0, st, a);
    return mpf_launch_status("k_pconv_x3_chunk");
}

int launch_pconv_x3_chunk(const MpfPConvArgs &a, hipStream_t st)
{
    if (a.nblk % 4 == 0) return launch_pconv_x3_chunk_nb<4>(a, st);
    if (a.nblk % 3 == 0) return launch_pconv_x3_chunk_nb<3>(a, st);
    if (a.nblk % 2 == 0) return launch_pconv_x3_chunk_nb<2>(a, st);
    return launch_pconv_x3_chunk_nb<1>(a, st);
}

int launch_pconv_x3_tile(const MpfPConvArgs &a, hipStream_t st)
{
    if (a.nblk == 1) return launch_pconv_x3_tile_nb<1>(a, st);
    if (a.nblk == 2) return launch_pconv_x3_tile_nb<2>(a, st);
    return launch_pconv_x3_tile_nb<3>(a, st);
}

inline size_t blocks_of(size_t n) { return (n + 255) / 256; }

}  // namespace

#define MPF_DTYPE_OK(d) ((d) == MPF_DTYPE_F32 || (d) == MPF_DTYPE_F64)
#define MPF_PCONV_DTYPE_OK(d) (MPF_DTYPE_OK(d) || (d) == MPF_DTYPE_F32X3 || (d) == MPF_DTYPE_F32X3_TILE || (d) == MPF_DTYPE_F32X3_CHUNK)

extern "C" int mpf_pconv(const MpfPConvArgs *args, void *stream)
{
    MPF_REQUIRE(args != nullptr, "mpf_pconv: null argument block");
    const MpfPConvArgs &a = *args;
    MPF_REQUIRE(MPF_PCONV_DTYPE_OK(a.dtype), "mpf_pconv: dtype must be MPF_DTYPE_F32, MPF_DTYPE_F64 or one of the MPF_DTYPE_F32X3 forms");
    MPF_REQUIRE(a.srcA && a.wpack && a.out, "mpf_pconv: null source / weights / output");
    MPF_REQUIRE(a.ksize == 1 || a.ksize == 3 || a.ksize == 7, "mpf_pconv: kernel size must be 1, 3 or 7");
    MPF_REQUIRE((a.stride == 1 || a.stride == 2) && a.pad >= 0 && a.pad <= a.ksize / 2 && (a.up == 0 || a.up == 1 || a.up == 2), "mpf_pconv: bad stride / padding / upsampling");
    MPF_REQUIRE(a.up != 2 || (a.dtype == MPF_DTYPE_F32X3_TILE && a.ksize == 3 && a.stride == 1 && a.pad == 1 && a.pad_mode == 1 && a.CB == 0 && (size_t)a.S * a.nblk * 4 <= 65535),
                "mpf_pconv: up = 2 (phase-decomposed x2 nearest) is the tile form's, 3 x 3 / stride 1 / reflection padding, one source");
    MPF_REQUIRE(a.pad_mode == 0 || (a.pad_mode == 1 && a.pad == 1 && a.Hin >= 2 && a.Win >= 2), "mpf_pconv: reflection padding is pad 1 on at least 2 rows and columns");
    MPF_REQUIRE(a.S >= 1 && a.S <= 65535 && a.Hin >= 1 && a.Win >= 1, "mpf_pconv: bad shape");
    MPF_REQUIRE(a.CA >= 4 && a.CA % 4 == 0 && a.CB >= 0 && a.CB % 4 == 0 && (a.CB == 0 || a.srcB), "mpf_pconv: channel counts must be multiples of 4 (zero-padded)");
    MPF_REQUIRE(a.HA == (a.Hin >> (a.up ? 1 : 0)) && a.WA == (a.Win >> (a.up ? 1 : 0)) && (a.up == 0 || (a.Hin % 2 == 0 && a.Win % 2 == 0)), "mpf_pconv: source A does not match the input size");
    MPF_REQUIRE(a.Hout == (a.Hin + 2 * a.pad - a.ksize) / a.stride + 1 && a.Wout == (a.Win + 2 * a.pad - a.ksize) / a.stride + 1 && a.Hout >= 1 && a.Wout >= 1,
                "mpf_pconv: output size does not match the convolution");
    MPF_REQUIRE(a.nblk >= 1 && (a.nblk + 1) / 2 <= 65535, "mpf_pconv: bad row-block count");
    MPF_REQUIRE(a.epi >= EP_AFFINE && a.epi <= EP_GATED_PLANAR, "mpf_pconv: unknown epilogue");
    const bool gated = a.epi == EP_GATED || a.epi == EP_GATED_PLANAR;
    MPF_REQUIRE(gated ? a.bias != nullptr : (a.scale && a.shift), "mpf_pconv: missing epilogue rows");
    MPF_REQUIRE(a.epi != EP_GATED || (a.scale && a.shift), "mpf_pconv: the gated block needs its BatchNorm rows");
    MPF_REQUIRE(a.Cst >= 1 && (a.epi == EP_GATED_PLANAR || a.epi == EP_AFFINE_MAP || a.Cst % 4 == 0), "mpf_pconv: stored channel count must be a multiple of 4");
    MPF_REQUIRE(a.Cst <= a.nblk * (gated ? 8 : 16), "mpf_pconv: more stored channels than packed rows");
    MPF_REQUIRE(a.act >= 0 && a.act <= 2, "mpf_pconv: activation must be 0 (none), 1 (ReLU) or 2 (leaky ReLU)");
    MPF_REQUIRE(a.residual == nullptr || a.epi == EP_AFFINE, "mpf_pconv: a residual goes with the affine epilogue");
    const size_t esz = a.dtype == MPF_DTYPE_F64 ? 8 : 4;
    MPF_REQUIRE((size_t)a.HA * a.WA * a.CA * esz < 0x7FFFFFFFull && (size_t)a.Hin * a.Win * (a.CB ? a.CB : 4) * esz < 0x7FFFFFFFull && (size_t)a.Hout * a.Wout < 0x7FFFFFFFull / 64,
                "mpf_pconv: plane too large (2 GiB of one source per plane: 32-bit buffer offsets)");
    const size_t al = a.dtype == MPF_DTYPE_F64 ? 31 : 15;
    MPF_REQUIRE((((uintptr_t)a.srcA | (uintptr_t)a.srcB | (uintptr_t)a.wpack | (uintptr_t)a.out | (uintptr_t)a.scale | (uintptr_t)a.shift | (uintptr_t)a.bias |
                  (uintptr_t)a.residual) & al) == 0, "mpf_pconv: buffers must be aligned to one 4-channel vector");
    if (a.dtype == MPF_DTYPE_F32X3_TILE) {
        MPF_REQUIRE(a.ksize == 3 && a.stride == 1 && a.pad == 1 && a.nblk <= 3 && a.CA + a.CB <= 56 && (size_t)a.S * a.nblk <= 65535,
                    "mpf_pconv: the tile form is 3 x 3, stride 1, padding 1, at most 56 input channels and 3 row blocks");
        return launch_pconv_x3_tile(a, (hipStream_t)stream);
    }
    if (a.dtype == MPF_DTYPE_F32X3_CHUNK) {
        MPF_REQUIRE(a.ksize == 3 && a.stride == 1 && a.pad == 1 && (size_t)a.S * a.nblk <= 65535,
                    "mpf_pconv: the chunk form is 3 x 3, stride 1, padding 1");
        return launch_pconv_x3_chunk(a, (hipStream_t)stream);
    }
    MPF_REQUIRE(a.dtype != MPF_DTYPE_F32X3 || a.ksize <= 3, "mpf_pconv: the split-bf16 kernels are 1 x 1 and 3 x 3 (tap table in LDS)");
    if (a.dtype == MPF_DTYPE_F32X3) return launch_pconv_x3<6>(a, (hipStream_t)stream);
    return a.dtype == MPF_DTYPE_F64 ? launch_pconv<double>(a, (hipStream_t)stream) : launch_pconv<float>(a, (hipStream_t)stream);
}

static inline bool pvec_aligned(const void *p, int dtype) { return (((uintptr_t)p) & (dtype == MPF_DTYPE_F64 ? 31u : 15u)) == 0; }

extern "C" int mpf_pfmn_input(const float *d_image_3HW, const float *d_disp_HW, const float *d_plane_vals, int S, int H, int W, void *d_out, int dtype, void *stream)
{
    MPF_REQUIRE(d_image_3HW && d_disp_HW && d_plane_vals && d_out && S >= 1 && H >= 1 && W >= 1 && MPF_DTYPE_OK(dtype), "mpf_pfmn_input: bad argument");
    const int N = H * W;
    if (dtype == MPF_DTYPE_F64) hipLaunchKernelGGL((k_fmn_input<double>), dim3(blocks_of((size_t)S * N)), dim3(256), 0, (hipStream_t)stream, d_image_3HW, d_disp_HW, d_plane_vals, S, N, (double *)d_out);
    else hipLaunchKernelGGL((k_fmn_input<float>), dim3(blocks_of((size_t)S * N)), dim3(256), 0, (hipStream_t)stream, d_image_3HW, d_disp_HW, d_plane_vals, S, N, (float *)d_out);
    return mpf_launch_status("k_fmn_input");
}

extern "C" int mpf_pencoder_input(const float *d_image_3HW, const float *d_disp_HW, int H, int W, void *d_out, int dtype, void *stream)
{
    MPF_REQUIRE(d_image_3HW && d_disp_HW && d_out && H >= 1 && W >= 1 && MPF_DTYPE_OK(dtype), "mpf_pencoder_input: bad argument");
    const int N = H * W;
    if (dtype == MPF_DTYPE_F64) hipLaunchKernelGGL((k_encoder_input<double>), dim3(blocks_of(N)), dim3(256), 0, (hipStream_t)stream, d_image_3HW, d_disp_HW, N, (double *)d_out);
    else hipLaunchKernelGGL((k_encoder_input<float>), dim3(blocks_of(N)), dim3(256), 0, (hipStream_t)stream, d_image_3HW, d_disp_HW, N, (float *)d_out);
    return mpf_launch_status("k_encoder_input");
}

extern "C" int mpf_pbilinear2x(const void *d_src, int S, int h, int w, int C, void *d_dst, int dtype, void *stream)
{
    MPF_REQUIRE(d_src && d_dst && S >= 1 && h >= 1 && w >= 1 && C >= 4 && C % 4 == 0 && MPF_DTYPE_OK(dtype), "mpf_pbilinear2x: bad argument");
    MPF_REQUIRE((size_t)h * w * C < 0x7FFFFFFFull / 4, "mpf_pbilinear2x: plane too large");
    MPF_REQUIRE(pvec_aligned(d_src, dtype) && pvec_aligned(d_dst, dtype), "mpf_pbilinear2x: buffers must be aligned to one 4-channel vector");
    const size_t n = (size_t)S * 4 * h * w * (C / 4);
    if (dtype == MPF_DTYPE_F64) hipLaunchKernelGGL((k_bilinear2x<double>), dim3(blocks_of(n)), dim3(256), 0, (hipStream_t)stream, (const double *)d_src, S, h, w, C / 4, (double *)d_dst);
    else hipLaunchKernelGGL((k_bilinear2x<float>), dim3(blocks_of(n)), dim3(256), 0, (hipStream_t)stream, (const float *)d_src, S, h, w, C / 4, (float *)d_dst);
    return mpf_launch_status("k_bilinear2x");
}

extern "C" int mpf_pper_plane(const void *d_feat_hwC, const void *d_cm, const void *d_fm, int S, int h, int w, int C, void *d_out, int dtype, void *stream)
{
    MPF_REQUIRE(d_feat_hwC && d_cm && d_fm && d_out && S >= 1 && h >= 1 && w >= 1 && C >= 4 && C % 4 == 0 && MPF_DTYPE_OK(dtype), "mpf_pper_plane: bad argument");
    MPF_REQUIRE(pvec_aligned(d_feat_hwC, dtype) && pvec_aligned(d_out, dtype), "mpf_pper_plane: buffers must be aligned to one 4-channel vector");
    const size_t n = (size_t)S * h * w * (C / 4 + 1);
    if (dtype == MPF_DTYPE_F64) hipLaunchKernelGGL((k_per_plane<double>), dim3(blocks_of(n)), dim3(256), 0, (hipStream_t)stream, (const double *)d_feat_hwC, (const double *)d_cm, (const double *)d_fm, S, h * w, C / 4, (double *)d_out);
    else hipLaunchKernelGGL((k_per_plane<float>), dim3(blocks_of(n)), dim3(256), 0, (hipStream_t)stream, (const float *)d_feat_hwC, (const float *)d_cm, (const float *)d_fm, S, h * w, C / 4, (float *)d_out);
    return mpf_launch_status("k_per_plane");
}

extern "C" int mpf_pplane_masks(const void *d_logits, int S, int H, int W, void *d_feature_mask, void *d_cum_mask, void *d_context_mask, void *const *d_cm, void *const *d_fm,
                                int dtype, void *stream)
{
    MPF_REQUIRE(d_logits && d_feature_mask && d_cum_mask && d_context_mask && d_cm && d_fm && MPF_DTYPE_OK(dtype), "mpf_pplane_masks: null pointer / bad dtype");
    MPF_REQUIRE(S > 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, "mpf_pplane_masks: H and W must be multiples of 32 (five x2 scales)");
    for (int i = 0; i < 5; ++i) MPF_REQUIRE(d_cm[i] && d_fm[i], "mpf_pplane_masks: null pyramid level %d", i);
    hipStream_t st = (hipStream_t)stream;
    const int N = H * W;
    if (dtype == MPF_DTYPE_F64) hipLaunchKernelGGL((k_softmax_planes<double>), dim3(blocks_of(N)), dim3(256), 0, st, (const double *)d_logits, S, N, (double *)d_feature_mask, (double *)d_cum_mask, (double *)d_context_mask);
    else hipLaunchKernelGGL((k_softmax_planes<float>), dim3(blocks_of(N)), dim3(256), 0, st, (const float *)d_logits, S, N, (float *)d_feature_mask, (float *)d_cum_mask, (float *)d_context_mask);
    for (int i = 0; i < 5; ++i) {
        const int k = 2 << i;
        const size_t n = (size_t)S * (H / k) * (W / k);
        if (dtype == MPF_DTYPE_F64) {
            hipLaunchKernelGGL((k_avgpool<double>), dim3(blocks_of(n)), dim3(256), 0, st, (const double *)d_context_mask, S, H, W, k, (double *)d_cm[i]);
            hipLaunchKernelGGL((k_avgpool<double>), dim3(blocks_of(n)), dim3(256), 0, st, (const double *)d_feature_mask, S, H, W, k, (double *)d_fm[i]);
        } else {
            hipLaunchKernelGGL((k_avgpool<float>), dim3(blocks_of(n)), dim3(256), 0, st, (const float *)d_context_mask, S, H, W, k, (float *)d_cm[i]);
            hipLaunchKernelGGL((k_avgpool<float>), dim3(blocks_of(n)), dim3(256), 0, st, (const float *)d_feature_mask, S, H, W, k, (float *)d_fm[i]);
        }
    }
    return mpf_launch_status("k_softmax_planes / k_avgpool");
}

extern "C" int mpf_pmaxpool3x3s2(const void *d_src_HWC, int Hin, int Win, int C, void *d_out, int dtype, void *stream)
{
    MPF_REQUIRE(d_src_HWC && d_out && Hin >= 1 && Win >= 1 && C >= 4 && C % 4 == 0 && MPF_DTYPE_OK(dtype), "mpf_pmaxpool3x3s2: bad argument");
    MPF_REQUIRE((size_t)Hin * Win * C < 0x7FFFFFFFull, "mpf_pmaxpool3x3s2: tensor too large");
    MPF_REQUIRE(pvec_aligned(d_src_HWC, dtype) && pvec_aligned(d_out, dtype), "mpf_pmaxpool3x3s2: buffers must be aligned to one 4-channel vector");
    const int Hout = (Hin - 1) / 2 + 1, Wout = (Win - 1) / 2 + 1, V = C / 4, n = Hout * Wout * V;
    if (dtype == MPF_DTYPE_F64) hipLaunchKernelGGL((k_maxpool3x3s2<double>), dim3(blocks_of(n)), dim3(256), 0, (hipStream_t)stream, (const double *)d_src_HWC, Hin, Win, V, Hout, Wout, (double *)d_out);
    else hipLaunchKernelGGL((k_maxpool3x3s2<float>), dim3(blocks_of(n)), dim3(256), 0, (hipStream_t)stream, (const float *)d_src_HWC, Hin, Win, V, Hout, Wout, (float *)d_out);
    return mpf_launch_status("k_maxpool3x3s2");
}
