// mpf_conv.hip - 3x3 convolution engine of the MPI producer network for gfx950.
//
// What it replaces: the 3x3 convolutions the reference runs over S plane-images at once - the feature-mask UNet
// (model/CPN/unet.py:18-69) and the gated-convolution decoder (model/CPN/decoder.py:10-71, :124-174) - together with the
// tensor plumbing around them (expand/cat of the per-plane inputs, reflection padding, x2 upsampling, skip concatenation,
// BatchNorm + activation, the gate product).  The reference's own GPU run keeps this network in half precision
// (gen_3dphoto_dynamic_v2.py:46,59,82-84); the engine does the same: fp16 storage, fp16 MFMA, fp32 accumulation,
// fp32 epilogue.
//
// Formulation: implicit GEMM  out[cout, pixel] = sum_{tap, cin} W[cout, tap, cin] * in[pixel + tap, cin]  on
// v_mfma_f32_16x16x32_f16 with the WEIGHTS as the A operand (16 output channels x 32 k) and 16 consecutive pixels of an
// image row as the B operand (32 k x 16 pixels).  Activations are NHWC fp16 with the channel count padded to a multiple
// of 8, so one lane's B fragment (8 consecutive k of one pixel) is ONE 16-byte LDS read and its 4 accumulators (4
// consecutive output channels of one pixel) are ONE 8-byte NHWC store - no transposes anywhere.
//   * k-steps: CT channels per tap are staged per chunk (CT = 8, 16 or 32); a 32-wide k-step covers 32/CT taps, so a
//     16-channel layer needs 5 MFMA steps instead of 9 and the 5-channel input layer 3.
//   * a workgroup (4 waves) stages the (TH*ST+2) x (TW*ST+2) x CT input tile of one plane in LDS ONCE per chunk through
//     a LOADER that synthesises the layer's virtual input on the fly (per-plane constant channels, x2 bilinear / nearest
//     upsampling, skip concatenation, shared encoder features gated by the per-plane context mask), so none of those
//     tensors is ever materialised in HBM.  The LDS pixel stride is padded per (CT, stride) so that every ds_read_b128
//     of a B fragment is bank-conflict free (strides found by enumeration of the 16-lane service groups).
//   * each wave owns PG pixel groups x NB output-channel blocks of accumulators; A fragments come pre-swizzled from
//     global memory (host-packed in fragment order: one coalesced 1 KB load per fragment, L1/L2 resident).
//   * the EPILOGUE applies bias/BatchNorm/activation or the gate product in fp32 registers.
#include "mpf_common.h"
#include <hip/hip_fp16.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int LD_FMN_INPUT = MPF_CONV_LD_FMN_INPUT;
constexpr int LD_DIRECT = MPF_CONV_LD_DIRECT;
constexpr int LD_BILINEAR_CAT = MPF_CONV_LD_BILINEAR_CAT;
constexpr int LD_NEAREST_PLANE = MPF_CONV_LD_NEAREST_PLANE;
constexpr int EP_AFFINE_RELU = MPF_CONV_EP_AFFINE_RELU;
constexpr int EP_AFFINE_RELU_F32 = MPF_CONV_EP_AFFINE_RELU_F32;
constexpr int EP_GATED_ELU = MPF_CONV_EP_GATED_ELU;
constexpr int EP_GATED_PLANAR_F32 = MPF_CONV_EP_GATED_PLANAR_F32;

__host__ __device__ constexpr int pix_stride_bytes(int CT, int ST)
{
    // conflict-free strides for ds_read_b128 B-fragment reads (16-lane service groups of gfx950), by enumeration
    return CT == 8 ? 16 : CT == 16 ? (ST == 1 ? 32 : 48) : (ST == 1 ? 96 : 80);
}

__device__ __forceinline__ u32x4 zero4() { return u32x4{0u, 0u, 0u, 0u}; }

__device__ __forceinline__ unsigned pack2(float a, float b)
{
    __half2 h = __floats2half2_rn(a, b);
    return *reinterpret_cast<unsigned *>(&h);
}

__device__ __forceinline__ void unpack8(const u32x4 &v, float *f)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned w = v[i];
        __half2 h = *reinterpret_cast<__half2 *>(&w);
        float2 t = __half22float2(h);
        f[2 * i] = t.x;
        f[2 * i + 1] = t.y;
    }
}

__device__ __forceinline__ u32x4 pack8(const float *f)
{
    return u32x4{pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
}

__device__ __forceinline__ int reflect_or_clamp(int i, int n, bool reflect)
{
    if (reflect) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// ---- loaders: 8 consecutive virtual input channels (vector vv) of plane s at conv-input pixel (y, x), as 8 fp16 -----------
template <int LOADER>
__device__ __forceinline__ u32x4 load_vec(const MpfConvArgs &a, int s, int y, int x, int vv)
{
    const bool reflect = a.pad_mode == 1;
    if (!reflect && (y < 0 || y >= a.Hin || x < 0 || x >= a.Win)) return zero4();
    y = reflect_or_clamp(y, a.Hin, reflect);
    x = reflect_or_clamp(x, a.Win, reflect);
    if constexpr (LOADER == LD_FMN_INPUT) {
        // (r, g, b, disparity map, plane disparity, 0, 0, 0)    model/CPN/unet.py:44-50
        if (vv != 0) return zero4();
        const float *img = (const float *)a.srcA;
        const float *dsp = (const float *)a.srcB;
        const size_t n = (size_t)a.Hin * a.Win, o = (size_t)y * a.Win + x;
        const float pd = a.plane_vals[s];
        return u32x4{pack2(img[o], img[n + o]), pack2(img[2 * n + o], dsp[o]), pack2(pd, 0.f), 0u};
    } else if constexpr (LOADER == LD_DIRECT) {
        if (vv * 8 >= a.CA) return zero4();
        const u32x4 *p = (const u32x4 *)a.srcA;
        return p[(((size_t)s * a.Hin + y) * a.Win + x) * (a.CA >> 3) + vv];
    } else if constexpr (LOADER == LD_BILINEAR_CAT) {
        const int va = a.CA >> 3;
        if (vv < va) {
            // x2 bilinear, align_corners=True (nn.Upsample in model/CPN/unet.py:42): src = dst * (in-1)/(out-1)
            const float fy = a.fparams[0] * (float)y, fx = a.fparams[1] * (float)x;
            int y0 = (int)fy, x0 = (int)fx;
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const int y1 = y0 + (y0 < a.HA - 1), x1 = x0 + (x0 < a.WA - 1);
            const u32x4 *p = (const u32x4 *)a.srcA + (size_t)s * a.HA * a.WA * va + vv;
            float v00[8], v01[8], v10[8], v11[8], o[8];
            unpack8(p[((size_t)y0 * a.WA + x0) * va], v00);
            unpack8(p[((size_t)y0 * a.WA + x1) * va], v01);
            unpack8(p[((size_t)y1 * a.WA + x0) * va], v10);
            unpack8(p[((size_t)y1 * a.WA + x1) * va], v11);
            const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = hy * (hx * v00[i] + lx * v01[i]) + ly * (hx * v10[i] + lx * v11[i]);
            return pack8(o);
        }
        vv -= va;
        if (vv * 8 >= a.CB) return zero4();
        const u32x4 *p = (const u32x4 *)a.srcB;
        return p[(((size_t)s * a.Hin + y) * a.Win + x) * (a.CB >> 3) + vv];
    } else {
        // LD_NEAREST_PLANE: [x2 nearest upsample of srcA (CA may be 0)] ++ [shared features * context mask, context mask,
        // feature mask] (model/CPN/decoder.py:131-150: the per-plane expansion of an encoder feature map)
        const int va = a.CA >> 3;
        if (vv < va) {
            const u32x4 *p = (const u32x4 *)a.srcA;
            const int ya = a.HA == a.Hin ? y : (y >> 1), xa = a.HA == a.Hin ? x : (x >> 1);
            return p[(((size_t)s * a.HA + ya) * a.WA + xa) * va + vv];
        }
        vv -= va;
        if (vv * 8 >= a.CB) return zero4();
        const size_t o = (size_t)y * a.Win + x, n = (size_t)a.Hin * a.Win;
        const float cm = a.cm[s * n + o];
        const int cf = a.CB - 8;                  // feature channels (multiple of 8); the last vector holds the two masks
        if (vv * 8 < cf) {
            const u32x4 *p = (const u32x4 *)a.srcB;
            float f[8];
            unpack8(p[o * (cf >> 3) + vv], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] *= cm;
            return pack8(f);
        }
        return u32x4{pack2(cm, a.fm[s * n + o]), 0u, 0u, 0u};
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- the kernel ------------------------------------------------------------------------------------------------------------
template <int ST, int CT, int LOADER, int EPI, int NB, int TH, int TW>
__global__ __launch_bounds__(256) void k_conv3x3(const MpfConvArgs a)
{
    constexpr int GROUPS = TH * TW / 16, PG = GROUPS / 4, GPR = TW / 16;
    constexpr int LW = TW * ST + 2, LH = TH * ST + 2, PIXB = pix_stride_bytes(CT, ST), VPP = CT / 8;
    constexpr int KS = (9 * CT + 31) / 32, TPS = 32 / CT;      // k-steps per chunk, taps per k-step
    static_assert(GROUPS % 4 == 0, "tile must give every wave the same number of pixel groups");
    __shared__ __attribute__((aligned(16))) unsigned char tile[LH * LW * PIXB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.z / a.ncg, cg = blockIdx.z - s * a.ncg;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const int ix0 = ox0 * ST - 1, iy0 = oy0 * ST - 1;

    f32x4 acc[PG][NB];
#pragma unroll
    for (int g = 0; g < PG; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[g][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q = lane >> 4, pi = lane & 15;
    const u32x4 *wp = (const u32x4 *)a.wpack + ((size_t)cg * NB) * 64 + lane;
    const int wstride = a.nblk * 64;                         // fragments of one k-step

    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        if (chunk) __syncthreads();
        for (int i = tid; i < LH * LW * VPP; i += 256) {
            const int p = i / VPP, v = i - p * VPP;
            const int ly = p / LW, lx = p - ly * LW;
            u32x4 val = load_vec<LOADER>(a, s, iy0 + ly, ix0 + lx, chunk * VPP + v);
            *reinterpret_cast<u32x4 *>(tile + p * PIXB + v * 16) = val;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            int slot = ks * TPS + q / VPP;
            slot = slot > 8 ? 8 : slot;                      // zero weights there; any finite operand will do
            const int ky = slot / 3, kx = slot - ky * 3;
            const int choff = (q % VPP) * 16;
            h8 af[NB];
            const u32x4 *wk = wp + (size_t)(chunk * KS + ks) * wstride;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                u32x4 w = wk[b * 64];
                af[b] = *reinterpret_cast<h8 *>(&w);
            }
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                const int gi = wave * PG + g, gy = gi / GPR, gx = (gi - gy * GPR) * 16;
                const int addr = ((gy * ST + ky) * LW + (gx + pi) * ST + kx) * PIXB + choff;
                u32x4 bv = *reinterpret_cast<const u32x4 *>(tile + addr);
                h8 bf = *reinterpret_cast<h8 *>(&bv);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[b], bf, acc[g][b], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds rows 4q..4q+3 (output channels) of column pi (pixel) of every 16x16 block --------------------
    const int rows = a.nblk * 16;
    const float *ep0 = a.ep, *ep1 = a.ep + rows, *ep2 = a.ep + 2 * rows;
#pragma unroll
    for (int g = 0; g < PG; ++g) {
        const int gi = wave * PG + g, gy = gi / GPR, gx = (gi - gy * GPR) * 16;
        const int oy = oy0 + gy, ox = ox0 + gx + pi;
        if (oy >= a.Hout || ox >= a.Wout) continue;
        const size_t opix = ((size_t)s * a.Hout + oy) * a.Wout + ox;
        if constexpr (EPI == EP_AFFINE_RELU || EPI == EP_AFFINE_RELU_F32) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int r0 = (cg * NB + b) * 16 + 4 * q;
                float y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = fmaxf(acc[g][b][i] * ep0[r0 + i] + ep1[r0 + i], 0.f);
                if constexpr (EPI == EP_AFFINE_RELU) {
                    if (r0 < a.Cst) *reinterpret_cast<u32x2 *>((__half *)a.out + opix * a.Cst + r0) = u32x2{pack2(y[0], y[1]), pack2(y[2], y[3])};
                } else {
                    if (r0 == 0) ((float *)a.out)[opix] = y[0];       // single-channel fp32 map [S,H,W]
                }
            }
        } else {
            constexpr int NF = NB / 2;
#pragma unroll
            for (int b = 0; b < NF; ++b) {
                const int rf = (cg * NB + b) * 16 + 4 * q, rm = (cg * NB + NF + b) * 16 + 4 * q;
                const int c0 = (cg * NF + b) * 16 + 4 * q;
                float y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gsum = (acc[g][b][i] + ep0[rf + i]) * sigmoidf_(acc[g][NF + b][i] + ep0[rm + i]);
                    if constexpr (EPI == EP_GATED_ELU) {
                        const float t = gsum * ep1[rf + i] + ep2[rf + i];
                        y[i] = t > 0.f ? t : (__expf(t) - 1.f);
                    } else {
                        y[i] = gsum;
                    }
                }
                if constexpr (EPI == EP_GATED_ELU) {
                    if (c0 < a.Cst) *reinterpret_cast<u32x2 *>((__half *)a.out + opix * a.Cst + c0) = u32x2{pack2(y[0], y[1]), pack2(y[2], y[3])};
                } else {
                    if (c0 < a.Cst) {                               // planar fp32 [S, Cst, H, W], Cst <= 4 channels used
                        const size_t n = (size_t)a.Hout * a.Wout;
                        float *o = (float *)a.out + (size_t)s * a.Cst * n + (size_t)oy * a.Wout + ox;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (c0 + i < a.Cst) o[(size_t)(c0 + i) * n] = y[i];
                    }
                }
            }
        }
    }
}

template <int ST, int CT, int LOADER, int EPI, int NB, int TH, int TW>
int launch(const MpfConvArgs &a, hipStream_t st)
{
    dim3 grid((a.Wout + TW - 1) / TW, (a.Hout + TH - 1) / TH, a.S * a.ncg);
    hipLaunchKernelGGL((k_conv3x3<ST, CT, LOADER, EPI, NB, TH, TW>), grid, dim3(256), 0, st, a);
    return mpf_launch_status("k_conv3x3");
}

template <int ST, int CT, int LOADER, int EPI>
int dispatch_nb(const MpfConvArgs &a, int nb, hipStream_t st)
{
    constexpr bool gated = EPI == EP_GATED_ELU || EPI == EP_GATED_PLANAR_F32;
    constexpr int TH = ST == 1 ? 8 : 4, TW = 32;
    switch (nb) {
    case 1: if constexpr (!gated) return launch<ST, CT, LOADER, EPI, 1, TH, TW>(a, st); else break;
    case 2: return launch<ST, CT, LOADER, EPI, 2, TH, TW>(a, st);
    case 4: return launch<ST, CT, LOADER, EPI, 4, TH, TW>(a, st);
    case 6: if constexpr (gated) return launch<ST, CT, LOADER, EPI, 6, TH, TW>(a, st); else break;
    case 8: return launch<ST, CT, LOADER, EPI, 8, TH, TW>(a, st);
    }
    mpf_set_error("mpf_conv3x3_f16: %d blocks per workgroup not built for this loader/epilogue", nb);
    return MPF_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int mpf_conv3x3_f16(const MpfConvArgs *args, void *stream)
{
    MPF_REQUIRE(args != nullptr, "mpf_conv3x3_f16: null argument block");
    const MpfConvArgs &a = *args;
    hipStream_t st = (hipStream_t)stream;
    MPF_REQUIRE(a.S > 0 && a.Hin > 0 && a.Win > 0 && a.Hout > 0 && a.Wout > 0, "mpf_conv3x3_f16: bad shape");
    MPF_REQUIRE(a.stride == 1 || a.stride == 2, "mpf_conv3x3_f16: stride must be 1 or 2");
    MPF_REQUIRE(a.Hout == (a.Hin - 1) / a.stride + 1 && a.Wout == (a.Win - 1) / a.stride + 1, "mpf_conv3x3_f16: output size does not match a pad-1 3x3 convolution");
    MPF_REQUIRE(a.ct == 8 || a.ct == 16 || a.ct == 32, "mpf_conv3x3_f16: channels per tap must be 8, 16 or 32");
    MPF_REQUIRE(a.nchunk > 0 && a.ncg > 0 && a.nblk > 0 && a.nblk % a.ncg == 0, "mpf_conv3x3_f16: bad block partition");
    MPF_REQUIRE((a.CA & 7) == 0 && (a.CB & 7) == 0, "mpf_conv3x3_f16: channel counts must be padded to multiples of 8");
    MPF_REQUIRE(a.wpack && a.ep && a.out, "mpf_conv3x3_f16: null weights/epilogue/output");
    MPF_REQUIRE(a.pad_mode == 0 || (a.Hin >= 2 && a.Win >= 2), "mpf_conv3x3_f16: reflection padding needs at least 2 rows and columns");
    MPF_REQUIRE((size_t)a.S * a.ncg <= 65535, "mpf_conv3x3_f16: planes x channel groups exceeds the grid limit");
    const int nb = a.nblk / a.ncg;
    const int key = a.loader * 1000 + a.epi * 100 + a.ct * 1 + a.stride * 10000;
    switch (key) {
    // feature-mask UNet (zero padding)
    case 10000 + LD_FMN_INPUT * 1000 + EP_AFFINE_RELU * 100 + 8:      return dispatch_nb<1, 8, LD_FMN_INPUT, EP_AFFINE_RELU>(a, nb, st);
    case 20000 + LD_DIRECT * 1000 + EP_AFFINE_RELU * 100 + 16:        return dispatch_nb<2, 16, LD_DIRECT, EP_AFFINE_RELU>(a, nb, st);
    case 20000 + LD_DIRECT * 1000 + EP_AFFINE_RELU * 100 + 32:        return dispatch_nb<2, 32, LD_DIRECT, EP_AFFINE_RELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_AFFINE_RELU * 100 + 32:        return dispatch_nb<1, 32, LD_DIRECT, EP_AFFINE_RELU>(a, nb, st);
    case 10000 + LD_BILINEAR_CAT * 1000 + EP_AFFINE_RELU * 100 + 32:  return dispatch_nb<1, 32, LD_BILINEAR_CAT, EP_AFFINE_RELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_AFFINE_RELU_F32 * 100 + 16:    return dispatch_nb<1, 16, LD_DIRECT, EP_AFFINE_RELU_F32>(a, nb, st);
    // gated decoder (reflection padding)
    case 10000 + LD_NEAREST_PLANE * 1000 + EP_GATED_ELU * 100 + 32:   return dispatch_nb<1, 32, LD_NEAREST_PLANE, EP_GATED_ELU>(a, nb, st);
    case 10000 + LD_NEAREST_PLANE * 1000 + EP_GATED_ELU * 100 + 16:   return dispatch_nb<1, 16, LD_NEAREST_PLANE, EP_GATED_ELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_GATED_ELU * 100 + 32:          return dispatch_nb<1, 32, LD_DIRECT, EP_GATED_ELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_GATED_PLANAR_F32 * 100 + 16:   return dispatch_nb<1, 16, LD_DIRECT, EP_GATED_PLANAR_F32>(a, nb, st);
    }
    mpf_set_error("mpf_conv3x3_f16: combination loader=%d epilogue=%d ct=%d stride=%d is not built", a.loader, a.epi, a.ct, a.stride);
    return MPF_ERR_UNSUPPORTED;
}
