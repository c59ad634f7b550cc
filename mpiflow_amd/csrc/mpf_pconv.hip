// mpf_pconv.hip - the PARITY-GRADE engine of the MPI producer network: every convolution of
// MPIPredictor.forward (reference model/AdaMPI.py:55-78) - the RGBD ResNet-18 encoder (model/CPN/encoder.py:20-101), the
// feature-mask UNet (model/CPN/unet.py:18-69) and the gated decoder (model/CPN/decoder.py:10-71, :124-174) - in the arithmetic
// of the reference's CPU path: fp32 storage, fp32 products, fp32 accumulation (v_mfma_f32_16x16x4_f32) in blocks of 64 products
// whose partial sums are carried in fp64, or - same code, T = double - fp64 throughout (v_mfma_f64_16x16x4_f64), which is what
// the tests use to show that the engine computes the reference's network and not something 1e-4 away from it.
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
    // lane (m, g): acc[nb][pg][i] = LOGICAL row 4g + i of block bg * NB + nb, pixel p0 + 16 pg + m (the host permutes the rows of a block for the
    // fp64 instruction, whose C/D layout is row = g + 4 i: mpiflow_amd/model/precise.py)
    const T *scale = reinterpret_cast<const T *>(a.scale), *shift = reinterpret_cast<const T *>(a.shift);
    T *out = reinterpret_cast<T *>(a.out);
#pragma unroll
    for (int pg = 0; pg < PG; ++pg) {
        const int p = p0 + 16 * pg + m;
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

inline size_t blocks_of(size_t n) { return (n + 255) / 256; }

}  // namespace

#define MPF_DTYPE_OK(d) ((d) == MPF_DTYPE_F32 || (d) == MPF_DTYPE_F64)

extern "C" int mpf_pconv(const MpfPConvArgs *args, void *stream)
{
    MPF_REQUIRE(args != nullptr, "mpf_pconv: null argument block");
    const MpfPConvArgs &a = *args;
    MPF_REQUIRE(MPF_DTYPE_OK(a.dtype), "mpf_pconv: dtype must be MPF_DTYPE_F32 or MPF_DTYPE_F64");
    MPF_REQUIRE(a.srcA && a.wpack && a.out, "mpf_pconv: null source / weights / output");
    MPF_REQUIRE(a.ksize == 1 || a.ksize == 3 || a.ksize == 7, "mpf_pconv: kernel size must be 1, 3 or 7");
    MPF_REQUIRE((a.stride == 1 || a.stride == 2) && a.pad >= 0 && a.pad <= a.ksize / 2 && (a.up == 0 || a.up == 1), "mpf_pconv: bad stride / padding / upsampling");
    MPF_REQUIRE(a.pad_mode == 0 || (a.pad_mode == 1 && a.pad == 1 && a.Hin >= 2 && a.Win >= 2), "mpf_pconv: reflection padding is pad 1 on at least 2 rows and columns");
    MPF_REQUIRE(a.S >= 1 && a.S <= 65535 && a.Hin >= 1 && a.Win >= 1, "mpf_pconv: bad shape");
    MPF_REQUIRE(a.CA >= 4 && a.CA % 4 == 0 && a.CB >= 0 && a.CB % 4 == 0 && (a.CB == 0 || a.srcB), "mpf_pconv: channel counts must be multiples of 4 (zero-padded)");
    MPF_REQUIRE(a.HA == (a.Hin >> a.up) && a.WA == (a.Win >> a.up) && (a.up == 0 || (a.Hin % 2 == 0 && a.Win % 2 == 0)), "mpf_pconv: source A does not match the input size");
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
