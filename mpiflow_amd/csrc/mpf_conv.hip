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

// Timing ablations of the conv kernels (tools/build_ablate_conv.sh builds SEPARATE libraries with -DMPF_CONV_ABLATE=<bits>; results INVALID), compile-time so that
// the code around the removed part is scheduled as in the shipped kernel: 2 no MFMAs, 4 loader arithmetic replaced by a copy of one tap / one map, 8 no barriers
// in the chunk loop, 16 no global -> LDS copies (fragments, raw tile).  In the shipped build ABL() is constant false.
#ifdef MPF_CONV_ABLATE
#define ABL(bit) (((MPF_CONV_ABLATE) & (bit)) != 0)
#else
#define ABL(bit) false
#endif

namespace {

constexpr int LD_FMN_INPUT = MPF_CONV_LD_FMN_INPUT;
constexpr int LD_DIRECT = MPF_CONV_LD_DIRECT;
constexpr int LD_BILINEAR_CAT = MPF_CONV_LD_BILINEAR_CAT;
constexpr int LD_NEAREST_PLANE = MPF_CONV_LD_NEAREST_PLANE;
constexpr int LD_FMN_SYNTH = MPF_CONV_LD_FMN_SYNTH;
constexpr int LD_BILINEAR_SYNTH = MPF_CONV_LD_BILINEAR_SYNTH;
constexpr int LD_NEAREST_PHASE = MPF_CONV_LD_NEAREST_PHASE;
constexpr bool is_bilinear(int loader) { return loader == LD_BILINEAR_CAT || loader == LD_BILINEAR_SYNTH; }
constexpr int EP_AFFINE_RELU = MPF_CONV_EP_AFFINE_RELU;
constexpr int EP_AFFINE_RELU_F32 = MPF_CONV_EP_AFFINE_RELU_F32;
constexpr int EP_GATED_ELU = MPF_CONV_EP_GATED_ELU;
constexpr int EP_GATED_PLANAR_F32 = MPF_CONV_EP_GATED_PLANAR_F32;
constexpr int EP_GATED_PLANAR_F32_PAIRED = MPF_CONV_EP_GATED_PLANAR_F32_PAIRED;
constexpr int EP_GATED_ELU_PAIRED = MPF_CONV_EP_GATED_ELU_PAIRED;
constexpr int EP_AFFINE_F32_NHWC = MPF_CONV_EP_AFFINE_F32_NHWC;

__host__ __device__ constexpr int pix_stride_bytes(int CT, int ST)
{
    // conflict-free strides for ds_read_b128 B-fragment reads (16-lane service groups of gfx950), by enumeration
    return CT == 8 ? 16 : CT == 16 ? (ST == 1 ? 32 : 48) : (ST == 1 ? 96 : 80);
}

// floats of the epilogue rows a workgroup parks in LDS - none where those bytes would cost a resident workgroup (the LDS is handed
// out in 1280-byte granules on gfx950: the 8-block stride-2 layer went from 3 to 2 workgroups per CU over 1 KB of rows, +16 % time)
__host__ __device__ constexpr int lds_workgroups(int bytes) { return 163840 / ((bytes + 1279) / 1280 * 1280); }
__host__ __device__ constexpr int ep_lds_floats(int EPI, int NB, int other_lds_bytes)
{
    const int n = (EPI == EP_GATED_PLANAR_F32 || EPI == EP_GATED_PLANAR_F32_PAIRED) ? 0 : EPI == EP_GATED_ELU_PAIRED ? 2 * NB * 8 : (EPI == EP_GATED_ELU ? 2 * (NB / 2) * 16 : 2 * NB * 16);
    return lds_workgroups(other_lds_bytes + n * 4) == lds_workgroups(other_lds_bytes) ? n : 0;
}

__device__ __forceinline__ u32x4 zero4() { return u32x4{0u, 0u, 0u, 0u}; }

__device__ __forceinline__ unsigned pack2(float a, float b)
{
    __half2 h = __floats2half2_rn(a, b);
    return *reinterpret_cast<unsigned *>(&h);
}

__device__ __forceinline__ __half2 as_h2(unsigned w) { return *reinterpret_cast<__half2 *>(&w); }
__device__ __forceinline__ unsigned as_u32(__half2 h) { return *reinterpret_cast<unsigned *>(&h); }

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

// ---- loaders ---------------------------------------------------------------------------------------------------------------
// A thread stages the SAME 8-channel vector v of NI pixels of the tile for every chunk, so everything that depends on the
// pixel only - border handling, source pixel indices, interpolation weights, the plane's mask values - is computed once
// (Stage) and the per-chunk work is one or four 16-byte loads plus the arithmetic on them.
template <int LOADER>
struct Stage {
    unsigned ia;            // pixel index into source A (always a valid address, also for padding pixels)
    unsigned ib;            // pixel index into source B
    bool ok;                // false: zero-padding pixel (or beyond the tile) -> the staged vector is zero
};
template <>
struct Stage<LD_BILINEAR_CAT> {
    unsigned ia, ib;                    // ia: byte offset of the top-left corner (this thread's vector) in the RAW low-resolution LDS tile
    float w00, w01, w10, w11;
    bool ok;
};
template <>
struct Stage<LD_BILINEAR_SYNTH> : Stage<LD_BILINEAR_CAT> {};
template <>
struct Stage<LD_NEAREST_PLANE> {
    unsigned ia, ib;
    __half2 cm2;            // (cm, cm)
    unsigned masks;         // (cm, fm) as two fp16
    bool ok;
};

// RAW tile of the x2 bilinear loader: the low-resolution source pixels a conv tile can touch, staged in LDS once per chunk
// (RH x RW pixels; the scale (in-1)/(out-1) is < 1/2, so LH rows span at most LH/2 + 1 source rows plus the +1 neighbour)
__host__ __device__ constexpr int raw_rows(int LH) { return LH / 2 + 2; }
__host__ __device__ constexpr int raw_cols(int LW) { return LW / 2 + 2; }

template <int LOADER>
__device__ __forceinline__ void stage_init(Stage<LOADER> &st, const MpfConvArgs &a, int s, int y, int x, bool in_tile, int ry0 = 0, int rx0 = 0,
                                           int raw_pitch = 0, int vpp = 1, int sv = 0)
{
    const bool reflect = a.pad_mode == 1;
    st.ok = in_tile && (reflect || (y >= 0 && y < a.Hin && x >= 0 && x < a.Win));
    y = reflect_or_clamp(y, a.Hin, reflect);
    x = reflect_or_clamp(x, a.Win, reflect);
    // B' of the factorised first layer (the plane channel's share, BatchNorm-scale * conv(0, 0, 0, 0, 1)) depends on the pixel only through WHICH of the nine taps fall
    // inside the image: one of 3 x 3 border classes (top / inner / bottom row x left / inner / right column).  bprime_table: B' is that [3][3][16] table instead
    // of an [H, W, 16] map - the same values bit for bit, read from L1 instead of 64 bytes per pixel and plane from L2 (layer 2 is bound by its L2 requests: 9 GB at
    // 25 TB/s; profiles/r5/engine_l2_requests.txt).  The bilinear layer keeps the class in the top bits of its pixel index: a field of its own cost registers
    [[maybe_unused]] const unsigned bcls = (unsigned)((y == 0 ? 0 : (y == a.Hin - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == a.Win - 1 ? 2 : 1)));
    if constexpr (LOADER == LD_FMN_INPUT || LOADER == LD_FMN_SYNTH) {
        st.ia = (unsigned)(y * a.Win + x);
        if constexpr (LOADER == LD_FMN_SYNTH) st.ib = a.bprime_table ? bcls : st.ia;
    } else if constexpr (LOADER == LD_DIRECT) {
        st.ia = (unsigned)((s * a.Hin + y) * a.Win + x);
    } else if constexpr (is_bilinear(LOADER)) {
        // x2 bilinear, align_corners=True (nn.Upsample in model/CPN/unet.py:42): src = dst * (in-1)/(out-1)
        const float fy = a.fparams[0] * (float)y, fx = a.fparams[1] * (float)x;
        int y0 = (int)fy, x0 = (int)fx;
        y0 = y0 > a.HA - 1 ? a.HA - 1 : y0;
        x0 = x0 > a.WA - 1 ? a.WA - 1 : x0;
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        // the raw tile holds rows ry0.. and columns rx0.. clamped to the source, so the +1 neighbours exist there at the borders too
        st.ia = (unsigned)((((y0 - ry0) * raw_pitch + (x0 - rx0)) * vpp + sv) * 16);
        st.w00 = hy * hx, st.w01 = hy * lx, st.w10 = ly * hx, st.w11 = ly * lx;
        st.ib = LOADER == LD_BILINEAR_SYNTH ? (unsigned)(y * a.Win + x) : (unsigned)((s * a.Hin + y) * a.Win + x);
        if constexpr (LOADER == LD_BILINEAR_SYNTH) st.ib |= bcls << 28;      // pixel index into A' (bits 0..27: H * W < 2^27, checked by the launcher) | border class
    } else {
        const int ya = a.HA == a.Hin ? y : (y >> 1), xa = a.HA == a.Hin ? x : (x >> 1);
        st.ia = (unsigned)((s * a.HA + ya) * a.WA + xa);
        st.ib = (unsigned)(y * a.Win + x);
        st.cm2 = __float2half2_rn(0.f);                      // the plane's mask values are filled in by the kernel (batched loads)
        st.masks = 0u;
    }
}

__device__ __forceinline__ u32x4 select4(bool c, const u32x4 &v) { return u32x4{c ? v[0] : 0u, c ? v[1] : 0u, c ? v[2] : 0u, c ? v[3] : 0u}; }

// The feature-mask network's FIRST layer never materialised (model/CPN/unet.py:44-50): its 64 plane-images differ only in the constant plane
// channel d_s, and the layer is affine in it up to the ReLU - c1[s] = relu(A' + d_s * B') with A' = BN(conv(r, g, b, disparity, 0)) per image and
// B' = BN-scale * conv(0, 0, 0, 0, 1) per size, both fp32 [H,W,16].  The consumers of c1 (layer 2, and layer 8's skip input) synthesise the 8
// channels of vector v of pixel `pix` here: same fp32 value the layer-1 launch would have rounded to fp16 (up to the order of two roundings).
__device__ __forceinline__ u32x4 synth_c1(const float *__restrict__ A, const float *__restrict__ B, unsigned pix, unsigned pixb, unsigned v, float d)
{
    const float4 *pa = reinterpret_cast<const float4 *>(A + ((size_t)pix * 16 + v * 8)), *pb = reinterpret_cast<const float4 *>(B + ((size_t)pixb * 16 + v * 8));
    const float4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    const float r[8] = {fmaxf(fmaf(b0.x, d, a0.x), 0.f), fmaxf(fmaf(b0.y, d, a0.y), 0.f), fmaxf(fmaf(b0.z, d, a0.z), 0.f), fmaxf(fmaf(b0.w, d, a0.w), 0.f),
                        fmaxf(fmaf(b1.x, d, a1.x), 0.f), fmaxf(fmaf(b1.y, d, a1.y), 0.f), fmaxf(fmaf(b1.z, d, a1.z), 0.f), fmaxf(fmaf(b1.w, d, a1.w), 0.f)};
    return pack8(r);
}

// 8 consecutive virtual input channels (vector vv = chunk * VPP + sv) of the staged pixel, as 8 fp16.  Written without
// divergent branches (clamped addresses + selects) so that the loads of all NI passes of a chunk can be in flight together.
template <int LOADER, int VPP>
__device__ __forceinline__ u32x4 stage_load(const Stage<LOADER> &st, const MpfConvArgs &a, int s, int chunk, int sv, const unsigned char *raw = nullptr,
                                            int raw_dx = 0, int raw_dy = 0, bool cheap = false)
{
    const unsigned vv = (unsigned)(chunk * VPP + sv);
    if constexpr (LOADER == LD_FMN_INPUT) {
        // (r, g, b, disparity map, plane disparity, 0, 0, 0)    model/CPN/unet.py:44-50
        const float *img = (const float *)a.srcA;
        const float *dsp = (const float *)a.srcB;
        const unsigned n = (unsigned)(a.Hin * a.Win), o = st.ia;
        return select4(st.ok, u32x4{pack2(img[o], img[n + o]), pack2(img[2 * n + o], dsp[o]), pack2(a.plane_vals[s], 0.f), 0u});
    } else if constexpr (LOADER == LD_DIRECT) {
        const unsigned va = (unsigned)a.CA >> 3;
        const unsigned vc = vv < va ? vv : va - 1;
        return select4(st.ok && vv < va, ((const u32x4 *)a.srcA)[(size_t)st.ia * va + vc]);
    } else if constexpr (LOADER == LD_FMN_SYNTH) {
        return select4(st.ok && vv < 2u, synth_c1((const float *)a.srcA, (const float *)a.srcB, st.ia, st.ib, vv < 2u ? vv : 1u, a.plane_vals[s]));
    } else if constexpr (is_bilinear(LOADER)) {
        const unsigned va = (unsigned)a.CA >> 3, vb = (unsigned)a.CB >> 3;       // va % VPP == 0 (checked by the launcher)
        if ((unsigned)(chunk * VPP) < va) {                                       // uniform: the whole chunk is source A
            const unsigned char *p = raw + st.ia;
            if (cheap) return select4(st.ok, *reinterpret_cast<const u32x4 *>(p));      // (ablation build only)
            const u32x4 r00 = *reinterpret_cast<const u32x4 *>(p), r01 = *reinterpret_cast<const u32x4 *>(p + raw_dx);
            const u32x4 r10 = *reinterpret_cast<const u32x4 *>(p + raw_dy), r11 = *reinterpret_cast<const u32x4 *>(p + raw_dy + raw_dx);
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const __half2 h00 = as_h2(r00[i]), h01 = as_h2(r01[i]), h10 = as_h2(r10[i]), h11 = as_h2(r11[i]);
                const float lo = fmaf(__low2float(h11), st.w11, fmaf(__low2float(h10), st.w10, fmaf(__low2float(h01), st.w01, fmaf(__low2float(h00), st.w00, 0.f))));
                const float hi = fmaf(__high2float(h11), st.w11, fmaf(__high2float(h10), st.w10, fmaf(__high2float(h01), st.w01, fmaf(__high2float(h00), st.w00, 0.f))));
                o[i] = pack2(lo, hi);
            }
            return select4(st.ok, o);
        }
        const unsigned vq = vv - va, vc = vq < vb ? vq : vb - 1;
        if constexpr (LOADER == LD_BILINEAR_SYNTH) return select4(st.ok && vq < vb, synth_c1((const float *)a.srcB, a.cm, st.ib & 0x0fffffffu, a.bprime_table ? st.ib >> 28 : (st.ib & 0x0fffffffu), vc, a.plane_vals[s]));
        return select4(st.ok && vq < vb, ((const u32x4 *)a.srcB)[(size_t)st.ib * vb + vc]);
    } else {
        // [x2 nearest upsample of srcA (CA may be 0)] ++ [shared features * context mask, context mask, feature mask]
        // (model/CPN/decoder.py:131-150: the per-plane expansion of an encoder feature map)
        const unsigned va = (unsigned)a.CA >> 3, vb = (unsigned)a.CB >> 3, nf = vb ? vb - 1 : 0u;
        const bool isA = vv < va;
        const unsigned vq = vv - va;
        const bool isF = !isA && vq < nf, isM = !isA && vb && vq == nf;
        const u32x4 *pa = (const u32x4 *)a.srcA + ((size_t)st.ia * va + (isA ? vv : 0u));
        const u32x4 *pb = (const u32x4 *)a.srcB + ((size_t)st.ib * nf + (isF ? vq : 0u));
        const u32x4 *p = isA ? pa : pb;
        u32x4 f = zero4();
        if (isA || isF) f = *p;                              // neither: nothing to read (CA == 0 or CB == 0 leave a null base)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned g = as_u32(__hmul2(as_h2(f[i]), st.cm2));
            f[i] = isF ? g : f[i];
        }
        f[0] = isM ? st.masks : f[0];
        return select4(st.ok, f);
    }
}

// LDS-DMA issued by assembly: hipcc does not know of the copy, so it neither drains vmcnt in front of the LDS reads that follow (with the builtin it waits
// for every pending copy before ANY read of the array) nor moves memory operations across it ("memory").  16 bytes per lane to lds_base + lane * 16;
// lds_base is wave-uniform (an SGPR).  The consumer's wait is an explicit s_waitcnt vmcnt(0) in front of a barrier.
__device__ __forceinline__ void dma16_async(const void *gptr, unsigned lds_base)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_base) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// ---- the kernel ------------------------------------------------------------------------------------------------------------
// WALK: the workgroup walks a.pw CONSECUTIVE planes at its tile position (grid: S / pw plane groups).  Everything that depends on the pixel only - border
// handling, the bilinear weights and raw-tile offsets, tap offsets, the parked epilogue rows: about a third of the VALU work of a few-channel full-resolution
// layer - is computed once per workgroup instead of once per plane; per plane only the sources' plane offsets advance.  Same arithmetic per output.
template <int ST, int CT, int LOADER, int EPI, int NB, int TH, int TW, bool WLDS, bool WALK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB <= 2 ? 3 : (NB <= 6 ? 2 : 1))))
void k_conv3x3(const MpfConvArgs a, const int prefetch)
{
    constexpr int GROUPS = TH * TW / 16, PG = GROUPS / 4, GPR = TW / 16;
    constexpr int LW = TW * ST + 2, LH = TH * ST + 2, PIXB = pix_stride_bytes(CT, ST), VPP = CT / 8;
    constexpr int KS = (9 * CT + 31) / 32, TPS = 32 / CT;      // k-steps per chunk, taps per k-step
    constexpr int PPT = 256 / VPP, NI = (LH * LW + PPT - 1) / PPT;   // tile pixels staged per pass, passes
    constexpr int WVEC = KS * NB * 64, NW = (WVEC + 255) / 256;      // 16-byte weight vectors per chunk, per thread
    constexpr int TILE_BYTES = (LH * LW * PIXB + 255) / 256 * 256;
    static_assert(GROUPS % 4 == 0, "tile must give every wave the same number of pixel groups");
    constexpr bool RAW = is_bilinear(LOADER);
    constexpr int RH = raw_rows(LH), RW = raw_cols(LW), RAWVEC = RH * RW * VPP, NR = (RAWVEC + 255) / 256;
    constexpr int WL_BYTES = WLDS ? KS * NB * 1024 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // the walking form keeps TWO buffers of fragments and of the raw tile: the next step's copies are issued in front of this step's MFMA phase (PFB = 2)
    constexpr int PFB = (WALK && WLDS) ? 2 : 1;
    // (the one-block bilinear layer prefetches its raw tile only: with ONE 5 KB buffer of fragments, copied at the head of each step, l8s runs 0.905 ms against 0.925)
    constexpr int PFW = (NB == 1 && is_bilinear(LOADER)) ? 1 : PFB;      // buffers of fragments
    unsigned char *tile = lds, *wlds0 = lds + TILE_BYTES;     // input tile | this chunk's A fragments (shared by the 4 waves)
    unsigned char *raw0 = lds + TILE_BYTES + PFW * WL_BYTES;  // | raw low-resolution tile of the bilinear loader
    // | this workgroup's slice of the epilogue rows, parked at kernel entry: read from global memory in the
    // epilogue, hipcc sinks each block's loads into that block's store branch, i.e. 2 dependent L2 round trips per block with
    // nothing left to overlap them (profiles/r2/engine_epilogue_rows_in_lds.txt)
    constexpr int RAW_BYTES_ = RAW ? RH * RW * VPP * 16 : 0;
    float *eplds = reinterpret_cast<float *>(lds + TILE_BYTES + PFW * WL_BYTES + PFB * RAW_BYTES_);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // plane_major: the plane index is the FASTEST grid dimension, so the S workgroups of one tile are dispatched back to back (8 per XCD) and find the
    // per-image sources they share (LD_FMN_SYNTH / LD_BILINEAR_SYNTH: the A', B' maps) in that XCD's L2 instead of re-fetching them per plane
    const unsigned bx = a.plane_major ? blockIdx.y : blockIdx.x, by = a.plane_major ? blockIdx.z : blockIdx.y, bz = a.plane_major ? blockIdx.x : blockIdx.z;
    const int npw = WALK ? a.pw : 1;
    const bool pf = PFB == 2 && prefetch != 0 && a.nchunk > 1;          // prefetch the next step's fragments / raw tile (launcher: mpf_tune("conv_pf"))
    int step = 0;                                             // (plane, chunk) steps this workgroup has started
    const int sgrp = (int)bz / a.ncg, cg = (int)bz - sgrp * a.ncg;
    int s = sgrp * npw;                                       // first plane of this workgroup
    // affine epilogues use rows 0, 1 of all NB blocks; the gated one rows 1, 2 of its NB/2 feature blocks; the planar one none.
    // One value per thread, loaded here (the round trip overlaps the staging set-up) and parked in LDS: in its own region where
    // that costs no resident workgroup (EPW > 0), else in the input tile's space once the last MFMA phase is over.
    constexpr bool EP_GATED = EPI == EP_GATED_ELU || EPI == EP_GATED_PLANAR_F32 || EPI == EP_GATED_PLANAR_F32_PAIRED || EPI == EP_GATED_ELU_PAIRED;
    constexpr int EPN = (EPI == EP_GATED_PLANAR_F32 || EPI == EP_GATED_PLANAR_F32_PAIRED) ? 0 : EPI == EP_GATED_ELU_PAIRED ? NB * 8 : (EPI == EP_GATED_ELU ? (NB / 2) * 16 : NB * 16);
    constexpr int EPW = ep_lds_floats(EPI, NB, TILE_BYTES + PFW * WL_BYTES + PFB * RAW_BYTES_) / 2;
    static_assert(2 * EPN <= 256 && 2 * EPN * 4 <= TILE_BYTES, "one epilogue value per thread");
    float epv = 0.f;
    if (tid < 2 * EPN) {
        const int row = tid / (EPN ? EPN : 1), c = tid - row * EPN;
        epv = a.ep[(row + (EP_GATED ? 1 : 0)) * (a.nblk * 16) + cg * (NB * 16) + c];
    }
    const int ox0 = (int)bx * TW, oy0 = (int)by * TH;
    const int ix0 = ox0 * ST - 1, iy0 = oy0 * ST - 1;

    // staging slots of this thread: vector sv of tile pixels sp + k * PPT
    const int sv = tid % VPP, sp = tid / VPP;
    // bilinear loader: origin of the raw tile = source pixel of the tile's first (clamped) row / column
    int ry0 = 0, rx0 = 0;
    unsigned rawsrc[RAW ? NR : 1];
    if constexpr (RAW) {
        ry0 = (int)(a.fparams[0] * (float)max(iy0, 0));
        rx0 = (int)(a.fparams[1] * (float)max(ix0, 0));
#pragma unroll
        for (int k = 0; k < NR; ++k) {                        // this thread's raw vectors: (pixel j / VPP, vector j % VPP)
            const int j = tid + k * 256, rp = j / VPP, rr = rp / RW, rq = rp - rr * RW;
            rawsrc[k] = (unsigned)((s * a.HA + min(ry0 + rr, a.HA - 1)) * a.WA + min(rx0 + rq, a.WA - 1)) * ((unsigned)a.CA >> 3) + (unsigned)(j % VPP);
        }
    }
    Stage<LOADER> stage[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int p = sp + k * PPT;
        const int ly = p / LW, lx = p - ly * LW;
        stage_init<LOADER>(stage[k], a, s, iy0 + ly, ix0 + lx, p < LH * LW, ry0, rx0, RW, VPP, sv);
    }
    if (EPW && tid < 2 * EPN) eplds[tid] = epv;

    const int q = lane >> 4, pi = lane & 15;
    // per-lane LDS byte offset of the tap each k-step reads (tap-packed layers: the tap depends on the lane's k-quarter)
    int tapoff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        int slot = ks * TPS + q / VPP;
        slot = slot > 8 ? 8 : slot;                          // zero weights there; any finite operand will do
        const int ky = slot / 3, kx = slot - ky * 3;
        tapoff[ks] = (ky * LW + pi * ST + kx) * PIXB + (q % VPP) * 16;
    }
    // weights (host-packed in fragment order, 1 KB per fragment).  WLDS: the chunk's KS x NB fragments go through LDS once
    // per workgroup - per-wave fragment loads cost as much L1 time as the MFMAs they feed, which is what bounds the
    // many-chunk / many-block layers; the few-chunk, LDS-hungry layers are better off loading fragments per wave (!WLDS).
    const u32x4 *wbase = (const u32x4 *)a.wpack + (unsigned)(cg * NB) * 64u;
    const unsigned wstride = (unsigned)a.nblk * 64u;         // vectors of one k-step in global memory

#pragma nounroll
    for (int pw = 0; pw < npw; ++pw, ++s) {
    if (WALK && pw) {
        // the next plane at the same tile position: only the sources' plane offsets move
        if constexpr (RAW) {
            const unsigned dr = (unsigned)(a.HA * a.WA) * ((unsigned)a.CA >> 3);
#pragma unroll
            for (int k = 0; k < NR; ++k) rawsrc[k] += dr;
        }
        if constexpr (LOADER == LD_DIRECT || LOADER == LD_NEAREST_PLANE) {
            const unsigned da = LOADER == LD_DIRECT ? (unsigned)(a.Hin * a.Win) : (unsigned)(a.HA * a.WA);
#pragma unroll
            for (int k = 0; k < NI; ++k) stage[k].ia += da;
        }
        if constexpr (LOADER == LD_BILINEAR_CAT) {
            const unsigned db = (unsigned)(a.Hin * a.Win);
#pragma unroll
            for (int k = 0; k < NI; ++k) stage[k].ib += db;
        }
    }
    if constexpr (LOADER == LD_NEAREST_PLANE) {
        // the plane's mask values of the staged pixels: all 2 * NI loads first, conversions after (written per pixel, hipcc waited
        // for each load before issuing the next: 6 dependent round trips at the head of every workgroup)
        if (a.CB) {
            float cmv[NI], fmv[NI];
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const size_t o = (size_t)s * a.Hin * a.Win + stage[k].ib;
                cmv[k] = a.cm[o];
                fmv[k] = a.fm[o];
            }
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                stage[k].cm2 = __float2half2_rn(cmv[k]);
                stage[k].masks = pack2(cmv[k], fmv[k]);
            }
        }
    }
    // gated layers start their accumulators at the convolution biases (row 4q+i of block b), the others at zero
    f32x4 acc[PG][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EP_GATED) {
            const float *bias = a.ep + (cg * NB + b) * 16 + 4 * q;
            init = f32x4{bias[0], bias[1], bias[2], bias[3]};
        }
#pragma unroll
        for (int g = 0; g < PG; ++g) acc[g][b] = init;
    }

    for (int chunk = 0; chunk < a.nchunk; ++chunk, ++step) {
        const bool fetched = pf && step > 0;                   // this step's copies were issued in front of the previous step's MFMA phase
        unsigned char *wlds = wlds0 + (pf && PFW == 2 ? (step & 1) * WL_BYTES : 0), *raw = raw0 + (pf ? (step & 1) * RAW_BYTES_ : 0);
        const bool rawchunk = RAW && (unsigned)(chunk * VPP) < ((unsigned)a.CA >> 3);       // uniform: a chunk of the upsampled source
        if (fetched && rawchunk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the prefetched raw tile: the issuing wave's wait, in front of the barrier
        if ((chunk || (WALK && pw)) && !ABL(8)) __syncthreads();  // (a walked plane: the previous plane's MFMA reads and parked epilogue rows are done with the tile)
        u32x4 staged[NI];
        if (WLDS && !(fetched && PFW == 2) && !(WALK && pw && a.nchunk == 1) && !ABL(16)) {   // (a walked single-chunk layer: the fragments of the first plane are still there)
            // LDS-DMA (global_load_lds_dwordx4): the fragments are a plain copy (host-packed in fragment order), so they go global ->
            // LDS without passing through registers or ds_write; destination = wave-uniform base + lane * 16, i.e. one 1 KB fragment
            // per wave instruction.  The explicit vmcnt(0) + __syncthreads() below drains it, and the barrier
            // at the top of the loop keeps it from overtaking the previous chunk's fragment reads.  Against staging through
            // registers: -20 % on the full-resolution bilinear layer, -5..-10 % on the decoder's 24-block layers, and it makes
            // LDS staging the better choice on three more layers (profiles/r2/engine_glds_layers.txt)
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const unsigned vb = (unsigned)(__builtin_amdgcn_readfirstlane(wave) * 64 + j * 256);
                if (NW * 256 == WVEC || vb < WVEC) {
                    const unsigned v = vb + (unsigned)lane;
                    const unsigned ks = v / (NB * 64), r = v - ks * (NB * 64);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wbase + ((unsigned)(chunk * KS + ks) * wstride + r)),
                                                     (__attribute__((address_space(3))) void *)(wlds + vb * 16), 16, 0, 0);
                }
            }
        }
        if constexpr (RAW) {
            if (rawchunk && !fetched && !ABL(16)) {
#pragma unroll
                for (int k = 0; k < NR; ++k)                                    // also a plain copy: LDS-DMA, lane-linear destination
                    if (NR * 256 == RAWVEC || tid + k * 256 < RAWVEC)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const u32x4 *)a.srcA + (rawsrc[k] + (unsigned)(chunk * VPP))),
                                                         (__attribute__((address_space(3))) void *)(raw + (__builtin_amdgcn_readfirstlane(wave) * 64 + k * 256) * 16), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the raw tile's LDS-DMA: the issuing wave's wait, spelled out (see below)
                if (!ABL(8)) __syncthreads();
            }
        }
        if constexpr (RAW) {
            // the source of a chunk (upsampled A from the raw tile / skip tensor B) is uniform: branch ONCE around the NI passes, so that
            // the B loads of all passes are in flight together (inside stage_load each pass had its own branch and its own wait)
            if (rawchunk) {
#pragma unroll
                for (int k = 0; k < NI; ++k) staged[k] = stage_load<LOADER, VPP>(stage[k], a, s, chunk, sv, raw, VPP * 16, RW * VPP * 16, ABL(4));
            } else {
                const unsigned va = (unsigned)a.CA >> 3, vb = (unsigned)a.CB >> 3, vq = (unsigned)(chunk * VPP + sv) - va, vc = vq < vb ? vq : vb - 1;
                u32x4 ld[NI];
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    if constexpr (LOADER == LD_BILINEAR_SYNTH) ld[k] = ABL(4) ? ((const u32x4 *)a.srcB)[(size_t)(stage[k].ib & 0x0fffffffu) * 4 + vc] : synth_c1((const float *)a.srcB, a.cm, stage[k].ib & 0x0fffffffu, a.bprime_table ? stage[k].ib >> 28 : (stage[k].ib & 0x0fffffffu), vc, a.plane_vals[s]);
                    else ld[k] = ((const u32x4 *)a.srcB)[(size_t)stage[k].ib * vb + vc];
                }
#pragma unroll
                for (int k = 0; k < NI; ++k) staged[k] = select4(stage[k].ok && vq < vb, ld[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NI; ++k) staged[k] = stage_load<LOADER, VPP>(stage[k], a, s, chunk, sv, raw, VPP * 16, RW * VPP * 16);
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int p = sp + k * PPT;
            if (NI * PPT == LH * LW || p < LH * LW) *reinterpret_cast<u32x4 *>(tile + p * PIXB + sv * 16) = staged[k];
        }
        // An LDS-DMA copy is ordered for other waves' reads only by the ISSUING wave's vmcnt wait in front of a barrier.  __syncthreads()'s fence emits that wait
        // where the compiler believes a copy is pending; round 5 found it losing track of a copy issued under a wave-dependent condition (mpf_pconv.hip,
        // MPF_COPY_BARRIER) - the fragment copies above are issued under one (vb < WVEC) - so the wait is explicit here too (it was already emitted: no change in time).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!ABL(8)) __syncthreads();
        if constexpr (PFB == 2) {
            // The next step's plain copies - its fragments, and its raw tile when it is a chunk of the upsampled source - go into the OTHER buffers now and land
            // during this step's MFMA phase: buffer (step + 1) & 1 was last read in step - 1 (fragments: its MFMA phase, raw tile: its staging), and every wave
            // has passed this step's barriers since.  Issued by assembly (dma16_async) so that hipcc does not put a vmcnt(0) in front of the MFMA phase's reads.
            const int nchunk_n = chunk + 1 < a.nchunk ? chunk + 1 : 0;
            if (pf && (chunk + 1 < a.nchunk || pw + 1 < npw) && !ABL(16)) {
                const unsigned nb_w = lds_addr(wlds0 + ((step + 1) & 1) * WL_BYTES), nb_r = lds_addr(raw0 + ((step + 1) & 1) * RAW_BYTES_);
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    const unsigned vb = (unsigned)(__builtin_amdgcn_readfirstlane(wave) * 64 + j * 256);
                    if ((NW * 256 == WVEC || vb < WVEC) && PFW == 2) {
                        const unsigned v = vb + (unsigned)lane;
                        const unsigned ks = v / (NB * 64), r = v - ks * (NB * 64);
                        dma16_async(wbase + ((unsigned)(nchunk_n * KS + ks) * wstride + r), nb_w + vb * 16);
                    }
                }
                if constexpr (RAW) {
                    if ((unsigned)(nchunk_n * VPP) < ((unsigned)a.CA >> 3)) {
                        const unsigned dr = nchunk_n ? 0u : (unsigned)(a.HA * a.WA) * ((unsigned)a.CA >> 3);      // chunk 0 of the NEXT plane
#pragma unroll
                        for (int k = 0; k < NR; ++k)
                            if (NR * 256 == RAWVEC || tid + k * 256 < RAWVEC)
                                dma16_async((const u32x4 *)a.srcA + (rawsrc[k] + dr + (unsigned)(nchunk_n * VPP)), nb_r + (unsigned)(__builtin_amdgcn_readfirstlane(wave) * 64 + k * 256) * 16);
                    }
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            h8 af[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                u32x4 w;
                if constexpr (WLDS) w = *reinterpret_cast<const u32x4 *>(wlds + ((ks * NB + b) * 64 + lane) * 16);
                else w = wbase[(unsigned)(chunk * KS + ks) * wstride + (unsigned)(b * 64 + lane)];
                af[b] = *reinterpret_cast<h8 *>(&w);
            }
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                const int gi = wave * PG + g, gy = gi / GPR, gx = (gi - gy * GPR) * 16;
#ifdef MPF_CONV_ABLATE_LDS
                // timing ablation ONLY (separate build, results invalid): every k-step of a chunk re-uses the B fragment of its first one - the upper bound
                // of any formulation that feeds several MFMAs from one LDS read (kn2row / shift-accumulate): profiles/r5/engine_lds_ablation.txt
                u32x4 bv = *reinterpret_cast<const u32x4 *>(tile + (gy * ST * LW + gx * ST) * PIXB + tapoff[0]);
#else
                u32x4 bv = *reinterpret_cast<const u32x4 *>(tile + (gy * ST * LW + gx * ST) * PIXB + tapoff[ks]);
#endif
                h8 bf = *reinterpret_cast<h8 *>(&bv);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (!ABL(2)) acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[b], bf, acc[g][b], 0, 0, 0);
            }
        }
    }

    const float *eprows = eplds;
    if constexpr (EPW == 0 && EPN > 0) {
        __syncthreads();                                      // every wave is done reading the tile
        if (tid < 2 * EPN) reinterpret_cast<float *>(tile)[tid] = epv;
        __syncthreads();
        eprows = reinterpret_cast<const float *>(tile);
    }
#define MPF_EP_PIXEL(g) const int gi = wave * PG + (g), gy = gi / GPR, gx = (gi - gy * GPR) * 16; const int oy = oy0 + gy, ox = ox0 + gx + pi;
#include "mpf_conv_epilogue.inc"
#undef MPF_EP_PIXEL
    }   // planes of this workgroup
}


// ---- the x2-nearest layers, phase-decomposed ----------------------------------------------------------------------------------
// upconv(i, 1) of the decoder (model/CPN/decoder.py:19-20,156-162) convolves cat(nearest_x2(x), skip) under reflection padding.  On the upsampled part the
// 3x3 window of an output pixel covers only 2 x 2 DISTINCT low-resolution pixels, and which ones - and with which sums of the nine weights - depends only on
// the output pixel's phase (py, px) = (y & 1, x & 1):
//     py = 0: rows (y/2 - 1, y/2) with weights (w[0], w[1] + w[2]);   py = 1: rows (y/2, y/2 + 1) with (w[0] + w[1], w[2]);   columns alike
// and reflection padding of the upsampled map is CLAMPING of the low-resolution index (row -1 reflects to row 1 = low-resolution row 0; row Hin to Hin - 2 =
// low-resolution row HA - 1).  So the layer is FOUR 2x2 convolutions on the low-resolution map - host-summed weights, 4 taps instead of 9, a 6 x 18 tile
// instead of 10 x 34 staged per chunk - plus the ordinary 3x3 on the skip channels, accumulated in the same registers.
//   * a wave owns ONE phase: its four pixel groups are the tile's four low-resolution rows, 16 low-resolution columns each (output pixels two apart);
//     the upsampled chunks' A fragments are that phase's own (nothing to share between the waves: loaded per wave, one coalesced 1 KB read each),
//     the skip chunks' go through LDS once per workgroup as in k_conv3x3;
//   * the skip chunks' B fragments are 16 pixels TWO apart in the 10 x 34 tile: the pixel stride of the stride-2 layers keeps their ds_read_b128 conflict-free.
// Chunks: first the ceil(CA / CT) upsampled ones (virtual width padded to whole chunks), then the skip's.  wpack = [chunkA][phase][KSA][nblk][64][8] ++
// [chunkB][KS][nblk][64][8] (mpiflow_amd/model/engine.py: pack_weights_up).
template <int CT, int EPI, int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB <= 2 ? 3 : (NB <= 6 ? 2 : 1))))
void k_conv3x3_up(const MpfConvArgs a)
{
    constexpr int TH = 8, TW = 32, PG = 4, VPP = CT / 8;
    constexpr int LW = TW + 2, LH = TH + 2, PIXB = pix_stride_bytes(CT, 2);          // skip chunks: full-resolution tile, fragment lanes two pixels apart
    constexpr int LWA = TW / 2 + 2, LHA = TH / 2 + 2, PIXA = pix_stride_bytes(CT, 1); // upsampled chunks: low-resolution tile
    constexpr int KS = (9 * CT + 31) / 32, TPS = 32 / CT, KSA = (4 * CT + 31) / 32;
    constexpr int PPT = 256 / VPP, NI = (LH * LW + PPT - 1) / PPT, NIA = (LHA * LWA + PPT - 1) / PPT;
    constexpr int WVEC = KS * NB * 64, NW = (WVEC + 255) / 256;
    constexpr int TB = LH * LW * PIXB, TA = LHA * LWA * PIXA;
    constexpr int TILE_BYTES = ((TB > TA ? TB : TA) + 255) / 256 * 256;
    constexpr int WL_BYTES = KS * NB * 1024;
    constexpr bool EP_GATED = true;
    constexpr int EPN = EPI == EP_GATED_ELU_PAIRED ? NB * 8 : (NB / 2) * 16;
    static_assert(EPI == EP_GATED_ELU || EPI == EP_GATED_ELU_PAIRED, "the x2-nearest layers are the decoder's gated convolutions");
    static_assert(2 * EPN <= 256, "one epilogue value per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *tile = lds, *wlds = lds + TILE_BYTES;
    float *eplds = reinterpret_cast<float *>(lds + TILE_BYTES + WL_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int py = wave >> 1, px = wave & 1;                  // this wave's phase
    const int s = (int)blockIdx.z / a.ncg, cg = (int)blockIdx.z - s * a.ncg;
    if (tid < 2 * EPN) {
        const int row = tid / EPN, c = tid - row * EPN;
        eplds[tid] = a.ep[(row + 1) * (a.nblk * 16) + cg * (NB * 16) + c];
    }
    const int ox0 = (int)blockIdx.x * TW, oy0 = (int)blockIdx.y * TH;
    const int ix0 = ox0 - 1, iy0 = oy0 - 1, ax0 = (ox0 >> 1) - 1, ay0 = (oy0 >> 1) - 1;
    const int sv = tid % VPP, sp = tid / VPP;
    const unsigned va = (unsigned)a.CA >> 3, vb = (unsigned)a.CB >> 3, nf = vb ? vb - 1 : 0u;
    const int nchunkA = (a.CA + CT - 1) / CT;

    // staging slots: the low-resolution pixels of the upsampled chunks (clamped = reflection of the upsampled map) and the full-resolution ones of the skip chunks
    unsigned aidx[NIA];
#pragma unroll
    for (int k = 0; k < NIA; ++k) {
        const int p = sp + k * PPT, ly = p / LWA, lx = p - ly * LWA;
        const int ya = min(max(ay0 + ly, 0), a.HA - 1), xa = min(max(ax0 + lx, 0), a.WA - 1);
        aidx[k] = (unsigned)((s * a.HA + ya) * a.WA + xa);
    }
    Stage<LD_NEAREST_PLANE> stage[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int p = sp + k * PPT, ly = p / LW, lx = p - ly * LW;
        stage_init<LD_NEAREST_PLANE>(stage[k], a, s, iy0 + ly, ix0 + lx, p < LH * LW);
    }
    if (a.CB) {
        float cmv[NI], fmv[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const size_t o = (size_t)s * a.Hin * a.Win + stage[k].ib;
            cmv[k] = a.cm[o];
            fmv[k] = a.fm[o];
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            stage[k].cm2 = __float2half2_rn(cmv[k]);
            stage[k].masks = pack2(cmv[k], fmv[k]);
        }
    }
    const int q = lane >> 4, pi = lane & 15;
    int tapA[KSA], tapB[KS];
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks) {
        int slot = ks * TPS + q / VPP;
        slot = slot > 3 ? 3 : slot;
        const int ty = slot >> 1, tx = slot & 1;
        tapA[ks] = ((py + ty) * LWA + pi + px + tx) * PIXA + (q % VPP) * 16;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        int slot = ks * TPS + q / VPP;
        slot = slot > 8 ? 8 : slot;
        const int ky = slot / 3, kx = slot - ky * 3;
        tapB[ks] = ((py + ky) * LW + 2 * pi + px + kx) * PIXB + (q % VPP) * 16;
    }
    const unsigned wstride = (unsigned)a.nblk * 64u;
    const u32x4 *wA = (const u32x4 *)a.wpack + (unsigned)(cg * NB) * 64u + (unsigned)lane;                 // + ((chunk * 4 + phase) * KSA + ks) * wstride + b * 64
    const u32x4 *wB = (const u32x4 *)a.wpack + (size_t)nchunkA * 4u * KSA * wstride + (unsigned)(cg * NB) * 64u;

    f32x4 acc[PG][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float *bias = a.ep + (cg * NB + b) * 16 + 4 * q;
        const f32x4 init = f32x4{bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
        for (int g = 0; g < PG; ++g) acc[g][b] = init;
    }

    // ---- upsampled chunks: 2x2 taps on the low-resolution tile, this wave's phase weights ----
    for (int chunk = 0; chunk < nchunkA; ++chunk) {
        if (chunk) __syncthreads();
        const unsigned vv = (unsigned)(chunk * VPP + sv);
        u32x4 st[NIA];
#pragma unroll
        for (int k = 0; k < NIA; ++k) st[k] = vv < va ? ((const u32x4 *)a.srcA)[(size_t)aidx[k] * va + vv] : zero4();
        const u32x4 *wk = wA + (unsigned)((chunk * 4 + wave) * KSA) * wstride;
        h8 af[KSA][NB];
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                u32x4 w = wk[(unsigned)ks * wstride + (unsigned)(b * 64)];
                af[ks][b] = *reinterpret_cast<h8 *>(&w);
            }
#pragma unroll
        for (int k = 0; k < NIA; ++k) {
            const int p = sp + k * PPT;
            if (NIA * PPT == LHA * LWA || p < LHA * LWA) *reinterpret_cast<u32x4 *>(tile + p * PIXA + sv * 16) = st[k];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks) {
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                u32x4 bv = *reinterpret_cast<const u32x4 *>(tile + g * LWA * PIXA + tapA[ks]);
                h8 bf = *reinterpret_cast<h8 *>(&bv);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][b], bf, acc[g][b], 0, 0, 0);
            }
        }
    }
    // ---- skip chunks: the ordinary 3x3 on the full-resolution tile (shared features x context mask, the two masks) ----
    for (int chunk = nchunkA; chunk < a.nchunk; ++chunk) {
        if (chunk) __syncthreads();
        const int cb = chunk - nchunkA;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const unsigned vbase = (unsigned)(__builtin_amdgcn_readfirstlane(wave) * 64 + j * 256);
            if (NW * 256 == WVEC || vbase < WVEC) {
                const unsigned v = vbase + (unsigned)lane;
                const unsigned ks = v / (NB * 64), r = v - ks * (NB * 64);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wB + ((unsigned)(cb * KS + ks) * wstride + r)),
                                                 (__attribute__((address_space(3))) void *)(wlds + vbase * 16), 16, 0, 0);
            }
        }
        const unsigned vq = (unsigned)(cb * VPP + sv);
        const bool isF = vq < nf, isM = vb && vq == nf;
        u32x4 st[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) st[k] = isF ? ((const u32x4 *)a.srcB)[(size_t)stage[k].ib * nf + vq] : zero4();
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            u32x4 f = st[k];
#pragma unroll
            for (int i = 0; i < 4; ++i) f[i] = as_u32(__hmul2(as_h2(f[i]), stage[k].cm2));
            f[0] = isM ? stage[k].masks : f[0];
            const int p = sp + k * PPT;
            if (NI * PPT == LH * LW || p < LH * LW) *reinterpret_cast<u32x4 *>(tile + p * PIXB + sv * 16) = select4(stage[k].ok, f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the fragments' LDS-DMA: the issuing wave's wait, in front of the barrier (see k_conv3x3)
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            h8 af[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                u32x4 w = *reinterpret_cast<const u32x4 *>(wlds + ((ks * NB + b) * 64 + lane) * 16);
                af[b] = *reinterpret_cast<h8 *>(&w);
            }
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                u32x4 bv = *reinterpret_cast<const u32x4 *>(tile + 2 * g * LW * PIXB + tapB[ks]);
                h8 bf = *reinterpret_cast<h8 *>(&bv);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[b], bf, acc[g][b], 0, 0, 0);
            }
        }
    }
    const float *eprows = eplds;
#define MPF_EP_PIXEL(g) const int oy = oy0 + 2 * (g) + py, ox = ox0 + 2 * pi + px;
#include "mpf_conv_epilogue.inc"
#undef MPF_EP_PIXEL
}

template <int CT, int EPI, int NB>
int launch_up(const MpfConvArgs &a, hipStream_t st)
{
    constexpr int TB = 10 * 34 * pix_stride_bytes(CT, 2), TA = 6 * 18 * pix_stride_bytes(CT, 1), KS = (9 * CT + 31) / 32;
    constexpr int EPN = EPI == EP_GATED_ELU_PAIRED ? NB * 8 : (NB / 2) * 16;
    constexpr int LDS_BYTES = ((TB > TA ? TB : TA) + 255) / 256 * 256 + KS * NB * 1024 + 2 * EPN * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "tile + weights exceed the LDS of a CU");
    static bool attr_set = false;
    if (!attr_set && LDS_BYTES > 64 * 1024) {
        MPF_HIP(hipFuncSetAttribute((const void *)k_conv3x3_up<CT, EPI, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    dim3 grid((a.Wout + 31) / 32, (a.Hout + 7) / 8, a.S * a.ncg);
    hipLaunchKernelGGL((k_conv3x3_up<CT, EPI, NB>), grid, dim3(256), LDS_BYTES, st, a);
    return mpf_launch_status("k_conv3x3_up");
}

// upconv(0, 1): the phase-decomposed x2-nearest layer WITHOUT a skip source and with a single chunk (CA <= 16 channels) - the whole convolution is two k-steps per pixel
// group.  A workgroup walks a.pw consecutive planes at its tile position: tap offsets, this wave's phase fragments (kept in registers), biases and epilogue rows are set up
// once; per plane ONE 16-byte load per thread (the next plane's is issued before this plane's MFMAs and lands during them and the epilogue), one LDS write into the other
// of two tile buffers, ONE barrier, 8 MFMAs per block and the gated epilogue.  Same arithmetic per output as k_conv3x3_up.
template <int EPI, int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))
void k_conv3x3_up1(const MpfConvArgs a)
{
    constexpr int CT = 16, PG = 4, VPP = 2, LWA = 18, LHA = 6, PIXA = pix_stride_bytes(CT, 1), KSA = 2, TPS = 2;
    constexpr int TILE_BYTES = (LHA * LWA * PIXA + 255) / 256 * 256;
    constexpr bool EP_GATED = true;
    constexpr int EPN = EPI == EP_GATED_ELU_PAIRED ? NB * 8 : (NB / 2) * 16;
    static_assert(EPI == EP_GATED_ELU || EPI == EP_GATED_ELU_PAIRED, "gated epilogues only");
    static_assert(PIXA == VPP * 16, "the tile is dense: pixel stride = the chunk's bytes");
    __shared__ __attribute__((aligned(16))) unsigned char tile2[2 * TILE_BYTES];
    __shared__ float eplds[2 * EPN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int py = wave >> 1, px = wave & 1;
    const int npw = a.pw > 1 ? a.pw : 1;
    const int sgrp = (int)blockIdx.z / a.ncg, cg = (int)blockIdx.z - sgrp * a.ncg;
    int s = sgrp * npw;
    if (tid < 2 * EPN) {
        const int row = tid / EPN, c = tid - row * EPN;
        eplds[tid] = a.ep[(row + 1) * (a.nblk * 16) + cg * (NB * 16) + c];
    }
    const int ox0 = (int)blockIdx.x * 32, oy0 = (int)blockIdx.y * 8;
    const int sv = tid % VPP, sp = tid / VPP;
    const unsigned va = (unsigned)a.CA >> 3;
    const bool stg = sp < LHA * LWA && (unsigned)sv < va;      // this thread stages vector sv of tile pixel sp (beyond the source's vectors: zeros, written once below)
    const int ly = sp / LWA, lx = sp - ly * LWA;
    const int ya = min(max((oy0 >> 1) - 1 + ly, 0), a.HA - 1), xa = min(max((ox0 >> 1) - 1 + lx, 0), a.WA - 1);
    const u32x4 *src = (const u32x4 *)a.srcA + ((size_t)((s * a.HA + ya) * a.WA + xa) * va + (unsigned)sv);
    const size_t dplane = (size_t)a.HA * a.WA * va;
    if (sp < LHA * LWA && !stg) {                             // padding vectors of both buffers: zero for the whole walk
        *reinterpret_cast<u32x4 *>(tile2 + sp * PIXA + sv * 16) = zero4();
        *reinterpret_cast<u32x4 *>(tile2 + TILE_BYTES + sp * PIXA + sv * 16) = zero4();
    }
    const int q = lane >> 4, pi = lane & 15;
    int tapA[KSA];
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks) {
        const int slot = ks * TPS + q / VPP, ty = slot >> 1, tx = slot & 1;
        tapA[ks] = ((py + ty) * LWA + pi + px + tx) * PIXA + (q % VPP) * 16;
    }
    const unsigned wstride = (unsigned)a.nblk * 64u;
    const u32x4 *wk = (const u32x4 *)a.wpack + (unsigned)(cg * NB) * 64u + (unsigned)lane + (unsigned)(wave * KSA) * wstride;
    h8 af[KSA][NB];
    f32x4 init[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks) {
            u32x4 w = wk[(unsigned)ks * wstride + (unsigned)(b * 64)];
            af[ks][b] = *reinterpret_cast<h8 *>(&w);
        }
        const float *bias = a.ep + (cg * NB + b) * 16 + 4 * q;
        init[b] = f32x4{bias[0], bias[1], bias[2], bias[3]};
    }
    u32x4 cur = zero4();
    if (stg) cur = *src;
    const float *eprows = eplds;
#pragma nounroll
    for (int pw = 0; pw < npw; ++pw, ++s) {
        unsigned char *tile = tile2 + (pw & 1) * TILE_BYTES;
        if (stg) *reinterpret_cast<u32x4 *>(tile + sp * PIXA + sv * 16) = cur;
        if (stg && pw + 1 < npw) { src += dplane; cur = *src; }
        __syncthreads();      // this plane's tile is complete; every wave is past the MFMA reads of the plane before the previous one (the buffer the NEXT plane writes)
        f32x4 acc[PG][NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < PG; ++g) acc[g][b] = init[b];
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks) {
#pragma unroll
            for (int g = 0; g < PG; ++g) {
                u32x4 bv = *reinterpret_cast<const u32x4 *>(tile + g * LWA * PIXA + tapA[ks]);
                h8 bf = *reinterpret_cast<h8 *>(&bv);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[g][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks][b], bf, acc[g][b], 0, 0, 0);
            }
        }
#define MPF_EP_PIXEL(g) const int oy = oy0 + 2 * (g) + py, ox = ox0 + 2 * pi + px;
#include "mpf_conv_epilogue.inc"
#undef MPF_EP_PIXEL
    }
}

template <int EPI, int NB>
int launch_up1(const MpfConvArgs &a, hipStream_t st)
{
    const int pw = a.pw > 1 ? a.pw : 1;
    dim3 grid((a.Wout + 31) / 32, (a.Hout + 7) / 8, (a.S / pw) * a.ncg);
    hipLaunchKernelGGL((k_conv3x3_up1<EPI, NB>), grid, dim3(256), 0, st, a);
    return mpf_launch_status("k_conv3x3_up1");
}

int g_conv_pf = 1;                // mpf_tune("conv_pf", 0 | 1): the walking kernels prefetch the next step's fragments / raw tile (scheduling only, same results)

template <int ST, int CT, int LOADER, int EPI, int NB, int TH, int TW, bool WLDS, bool WALK = false>
int launch_w(const MpfConvArgs &a, hipStream_t st)
{
    constexpr int LW = TW * ST + 2, LH = TH * ST + 2, KS = (9 * CT + 31) / 32;
    constexpr int RAW_BYTES = is_bilinear(LOADER) ? raw_rows(LH) * raw_cols(LW) * (CT / 8) * 16 : 0;
    constexpr int PFB = (WALK && WLDS) ? 2 : 1;                // the walking form double-buffers the fragments and the raw tile (prefetch)
    constexpr int PFW = (NB == 1 && is_bilinear(LOADER)) ? 1 : PFB;
    constexpr int LDS_OTHER = (LH * LW * pix_stride_bytes(CT, ST) + 255) / 256 * 256 + PFW * (WLDS ? KS * NB * 1024 : 0) + PFB * RAW_BYTES;
    constexpr int LDS_BYTES = LDS_OTHER + ep_lds_floats(EPI, NB, LDS_OTHER) * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "tile + weights exceed the LDS of a CU");
    static bool attr_set = false;
    if (!attr_set && LDS_BYTES > 64 * 1024) {
        MPF_HIP(hipFuncSetAttribute((const void *)k_conv3x3<ST, CT, LOADER, EPI, NB, TH, TW, WLDS, WALK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const int groups = WALK ? a.S / a.pw : a.S;                // plane groups: a walking workgroup owns a.pw consecutive planes
    dim3 grid((a.Wout + TW - 1) / TW, (a.Hout + TH - 1) / TH, groups * a.ncg);
    if (a.plane_major) grid = dim3(groups * a.ncg, (a.Wout + TW - 1) / TW, (a.Hout + TH - 1) / TH);
    hipLaunchKernelGGL((k_conv3x3<ST, CT, LOADER, EPI, NB, TH, TW, WLDS, WALK>), grid, dim3(256), LDS_BYTES, st, a, g_conv_pf);
    return mpf_launch_status("k_conv3x3");
}

template <int ST, int CT, int LOADER, int EPI, int NB, int TH, int TW>
int launch(const MpfConvArgs &a, hipStream_t st)
{
    if constexpr (NB <= 2) {                                  // the walking form is built for the few-block (full-resolution) layers only
        if (a.pw > 1) return a.wlds ? launch_w<ST, CT, LOADER, EPI, NB, TH, TW, true, true>(a, st) : launch_w<ST, CT, LOADER, EPI, NB, TH, TW, false, true>(a, st);
    } else if (a.pw > 1) {
        mpf_set_error("mpf_conv3x3_f16: planes per workgroup (pw = %d) needs at most 2 blocks per workgroup, got %d", a.pw, NB);
        return MPF_ERR_UNSUPPORTED;
    }
    return a.wlds ? launch_w<ST, CT, LOADER, EPI, NB, TH, TW, true>(a, st) : launch_w<ST, CT, LOADER, EPI, NB, TH, TW, false>(a, st);
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

// ---- plane masks: softmax over the planes, cumulative / context masks, and their average-pooled pyramid -----------------------
// One 1024-thread workgroup owns a 32 x 32 pixel block of all S planes; a wave owns an 8 x 8 sub-block in Morton order, so
// the 2x2, 4x4 and 8x8 block sums are lane reductions (xor 1|2, 4|8, 16|32); the 16x16 and 32x32 sums are combined from the
// per-wave sums parked in LDS.  Pass 1 is an online softmax (running max and sum), pass 2 normalises, accumulates the
// cumulative mask and emits everything - the logits are read twice and nothing else is read.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

__global__ __launch_bounds__(1024) void k_plane_masks(const float *__restrict__ logits, int S, int H, int W, float *__restrict__ fmask,
                                                       float *__restrict__ cum, float *__restrict__ cm2, float *__restrict__ fm2,
                                                       float *__restrict__ cm4, float *__restrict__ fm4, float *__restrict__ cm8,
                                                       float *__restrict__ fm8, float *__restrict__ cm16, float *__restrict__ fm16,
                                                       float *__restrict__ cm32, float *__restrict__ fm32)
{
    extern __shared__ float wsum[];                          // [S][16 waves][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Morton order inside the wave: lane bits (x0 y0 x1 y1 x2 y2)
    const int lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4), ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const int wx = wave & 3, wy = wave >> 2;
    const int x = blockIdx.x * 32 + wx * 8 + lx, y = blockIdx.y * 32 + wy * 8 + ly;
    const bool in = x < W && y < H;
    const size_t n = (size_t)H * W, o = (size_t)(in ? y : 0) * W + (in ? x : 0);
    // the logits are read CH planes at a time, all CH loads in flight together: plane by plane (load, wait, use) the kernel was 2 * S
    // dependent memory round trips long and nothing else (0.28 ms at 64 x 384 x 1280); same operations in the same order
    constexpr int CH = 16;
    float mx = -INFINITY, sum = 0.f;
    for (int s0 = 0; s0 < S; s0 += CH) {
        float v[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = logits[(size_t)min(s0 + j, S - 1) * n + o];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (s0 + j < S) {
                const float m2 = fmaxf(mx, v[j]);
                sum = sum * __expf(mx - m2) + __expf(v[j] - m2);
                mx = m2;
            }
        }
    }
    const float inv = 1.f / sum;
    const int H2 = H >> 1, W2 = W >> 1, H4 = H >> 2, W4 = W >> 2, H8 = H >> 3, W8 = W >> 3;
    float run = 0.f;
    for (int s0 = 0; s0 < S; s0 += CH) {
      float lg[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) lg[j] = logits[(size_t)min(s0 + j, S - 1) * n + o];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int s = s0 + j;
        if (s >= S) break;
        const float p = __expf(lg[j] - mx) * inv;
        const float ctx = 1.f - run;                         // context mask: 1 - cumulative mask of the planes in front
        run += p;
        if (in) {
            cum[s * n + o] = run;
            if (fmask) fmask[s * n + o] = p;
        }
        float c = in ? ctx : 0.f, f = in ? p : 0.f;
        // 2x2 / 4x4 sums by DPP (VALU operand permutes) instead of ds_bpermute: the 12 shuffles per plane were LDS-pipe time
        c += dpp_f<0xB1>(c); f += dpp_f<0xB1>(f);            // quad_perm [1,0,3,2]: lane ^ 1
        c += dpp_f<0x4E>(c); f += dpp_f<0x4E>(f);            // quad_perm [2,3,0,1]: lane ^ 2
        if ((lane & 3) == 0 && in) {
            const size_t q = (size_t)s * H2 * W2 + (size_t)(y >> 1) * W2 + (x >> 1);
            cm2[q] = c * 0.25f; fm2[q] = f * 0.25f;
        }
        c += dpp_f<0x141>(c); f += dpp_f<0x141>(f);          // row_half_mirror: the other quad of the 8 lanes (all its lanes hold its sum)
        c += dpp_f<0x140>(c); f += dpp_f<0x140>(f);          // row_mirror: the other half of the 16 lanes
        if ((lane & 15) == 0 && in) {
            const size_t q = (size_t)s * H4 * W4 + (size_t)(y >> 2) * W4 + (x >> 2);
            cm4[q] = c * 0.0625f; fm4[q] = f * 0.0625f;
        }
        c += __shfl_xor(c, 16); f += __shfl_xor(f, 16);
        c += __shfl_xor(c, 32); f += __shfl_xor(f, 32);
        if (lane == 0) {
            if (in) {
                const size_t q = (size_t)s * H8 * W8 + (size_t)(y >> 3) * W8 + (x >> 3);
                cm8[q] = c * (1.f / 64.f); fm8[q] = f * (1.f / 64.f);
            }
            wsum[(s * 16 + wave) * 2] = c;
            wsum[(s * 16 + wave) * 2 + 1] = f;
        }
      }
    }
    __syncthreads();
    // 16x16 (4 per block) and 32x32 (1 per block) sums for every plane
    const int H16 = H >> 4, W16 = W >> 4, H32 = H >> 5, W32 = W >> 5;
    for (int i = tid; i < S * 4; i += 1024) {
        const int s = i >> 2, k = i & 3, kx = k & 1, ky = k >> 1;
        const int bx = blockIdx.x * 2 + kx, by = blockIdx.y * 2 + ky;
        float c = 0.f, f = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = (ky * 2 + (j >> 1)) * 4 + kx * 2 + (j & 1);
            c += wsum[(s * 16 + w) * 2];
            f += wsum[(s * 16 + w) * 2 + 1];
        }
        if (bx < W16 && by < H16) {
            const size_t q = (size_t)s * H16 * W16 + (size_t)by * W16 + bx;
            cm16[q] = c * (1.f / 256.f); fm16[q] = f * (1.f / 256.f);
        }
    }
    for (int s = tid; s < S; s += 1024) {
        float c = 0.f, f = 0.f;
        for (int w = 0; w < 16; ++w) {
            c += wsum[(s * 16 + w) * 2];
            f += wsum[(s * 16 + w) * 2 + 1];
        }
        if ((int)blockIdx.x < W32 && (int)blockIdx.y < H32) {
            const size_t q = (size_t)s * H32 * W32 + (size_t)blockIdx.y * W32 + blockIdx.x;
            cm32[q] = c * (1.f / 1024.f); fm32[q] = f * (1.f / 1024.f);
        }
    }
}

}  // namespace

void mpf_conv_set_prefetch(int v) { g_conv_pf = v != 0; }          // mpf_tune("conv_pf", v) (mpf_render.hip)

extern "C" int mpf_plane_masks(const float *d_logits, int S, int H, int W, float *d_feature_mask, float *d_cum_mask, float *const *d_cm,
                               float *const *d_fm, void *stream)
{
    MPF_REQUIRE(d_logits && d_cum_mask && d_cm && d_fm, "mpf_plane_masks: null pointer");
    MPF_REQUIRE(S > 0 && S <= 512 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, "mpf_plane_masks: H and W must be multiples of 32 (five x2 scales), 1 <= S <= 512");
    for (int i = 0; i < 5; ++i) MPF_REQUIRE(d_cm[i] && d_fm[i], "mpf_plane_masks: null pyramid level %d", i);
    dim3 grid(W / 32, H / 32);
    hipLaunchKernelGGL(k_plane_masks, grid, dim3(1024), (size_t)S * 16 * 2 * sizeof(float), (hipStream_t)stream, d_logits, S, H, W, d_feature_mask,
                       d_cum_mask, d_cm[0], d_fm[0], d_cm[1], d_fm[1], d_cm[2], d_fm[2], d_cm[3], d_fm[3], d_cm[4], d_fm[4]);
    return mpf_launch_status("k_plane_masks");
}

extern "C" int mpf_conv3x3_f16(const MpfConvArgs *args, void *stream)
{
    MPF_REQUIRE(args != nullptr, "mpf_conv3x3_f16: null argument block");
    const MpfConvArgs &a = *args;
    hipStream_t st = (hipStream_t)stream;
    MPF_REQUIRE(a.S > 0 && a.Hin > 0 && a.Win > 0 && a.Hout > 0 && a.Wout > 0, "mpf_conv3x3_f16: bad shape");
    MPF_REQUIRE(a.stride == 1 || a.stride == 2, "mpf_conv3x3_f16: stride must be 1 or 2");
    MPF_REQUIRE(a.pw >= 0 && (a.pw <= 1 || a.S % a.pw == 0), "mpf_conv3x3_f16: planes per workgroup (pw) must divide S");
    MPF_REQUIRE(a.Hout == (a.Hin - 1) / a.stride + 1 && a.Wout == (a.Win - 1) / a.stride + 1, "mpf_conv3x3_f16: output size does not match a pad-1 3x3 convolution");
    MPF_REQUIRE(a.ct == 8 || a.ct == 16 || a.ct == 32, "mpf_conv3x3_f16: channels per tap must be 8, 16 or 32");
    MPF_REQUIRE(a.nchunk > 0 && a.ncg > 0 && a.nblk > 0 && a.nblk % a.ncg == 0, "mpf_conv3x3_f16: bad block partition");
    MPF_REQUIRE((a.CA & 7) == 0 && (a.CB & 7) == 0, "mpf_conv3x3_f16: channel counts must be padded to multiples of 8");
    MPF_REQUIRE(a.wpack && a.ep && a.out, "mpf_conv3x3_f16: null weights/epilogue/output");
    MPF_REQUIRE(a.pad_mode == 0 || (a.Hin >= 2 && a.Win >= 2), "mpf_conv3x3_f16: reflection padding needs at least 2 rows and columns");
    MPF_REQUIRE(!is_bilinear(a.loader) || (a.CA % a.ct == 0 && a.CA > 0 && a.CB > 0), "mpf_conv3x3_f16: the upsampled source must fill whole chunks");
    MPF_REQUIRE((a.loader != LD_FMN_SYNTH && a.loader != LD_BILINEAR_SYNTH) || (a.plane_vals && a.srcA && a.srcB && (a.loader == LD_FMN_SYNTH ? a.CA == 16 : (a.CB == 16 && a.cm))),
                "mpf_conv3x3_f16: the synthesised first-layer source needs the two fp32 maps (16 channels) and the plane values");
    MPF_REQUIRE(!a.bprime_table || ((a.loader == LD_FMN_SYNTH || a.loader == LD_BILINEAR_SYNTH) && a.Hin >= 2 && a.Win >= 2 && a.pad_mode == 0),
                "mpf_conv3x3_f16: bprime_table needs a synthesising loader, zero padding and at least 2 x 2 pixels");
    MPF_REQUIRE(a.loader != LD_BILINEAR_SYNTH || (size_t)a.Hin * a.Win < (1u << 27), "mpf_conv3x3_f16: LD_BILINEAR_SYNTH packs the pixel index into 27 bits");
    MPF_REQUIRE(a.plane_major ? ((a.Wout + 31) / 32 <= 65535 && (a.Hout + 3) / 4 <= 65535) : (size_t)a.S * a.ncg <= 65535, "mpf_conv3x3_f16: grid dimension limit exceeded");
    const int nb = a.nblk / a.ncg;
    if (a.loader == LD_NEAREST_PHASE) {
        MPF_REQUIRE(a.stride == 1 && a.pad_mode == 1 && a.CA > 0 && a.srcA && a.Hin == 2 * a.HA && a.Win == 2 * a.WA, "mpf_conv3x3_f16: the phase-decomposed loader is the x2-nearest, reflection-padded layer");
        MPF_REQUIRE(a.CB == 0 || (a.srcB && a.cm && a.fm && a.CB >= 16), "mpf_conv3x3_f16: the skip source needs its features and both masks");
        MPF_REQUIRE(a.nchunk == (a.CA + a.ct - 1) / a.ct + (a.CB + a.ct - 1) / a.ct, "mpf_conv3x3_f16: chunk count of the phase-decomposed layer (whole chunks per source)");
        MPF_REQUIRE(!a.plane_major, "mpf_conv3x3_f16: the phase-decomposed kernels do not reorder the grid");
        if (a.CB == 0 && a.CA <= 16 && a.ct == 16 && a.epi == EP_GATED_ELU && nb == 2) return launch_up1<EP_GATED_ELU, 2>(a, st);      // upconv(0,1): one chunk, walks a.pw planes
        MPF_REQUIRE(a.pw <= 1, "mpf_conv3x3_f16: only the single-chunk phase-decomposed layer walks planes");
        if (a.epi == EP_GATED_ELU && a.ct == 32 && nb == 4) return launch_up<32, EP_GATED_ELU, 4>(a, st);
        if (a.epi == EP_GATED_ELU && a.ct == 16 && nb == 6) return launch_up<16, EP_GATED_ELU, 6>(a, st);
        if (a.epi == EP_GATED_ELU && a.ct == 16 && nb == 4) return launch_up<16, EP_GATED_ELU, 4>(a, st);
        if (a.epi == EP_GATED_ELU && a.ct == 16 && nb == 2) return launch_up<16, EP_GATED_ELU, 2>(a, st);
        if (a.epi == EP_GATED_ELU_PAIRED && a.ct == 16 && nb == 3) return launch_up<16, EP_GATED_ELU_PAIRED, 3>(a, st);
        mpf_set_error("mpf_conv3x3_f16: phase-decomposed layer with epilogue=%d ct=%d blocks=%d is not built", a.epi, a.ct, nb);
        return MPF_ERR_UNSUPPORTED;
    }
    const int key = a.loader * 1000 + a.epi * 100 + a.ct * 1 + a.stride * 10000;
    switch (key) {
    // feature-mask UNet (zero padding)
    case 10000 + LD_FMN_INPUT * 1000 + EP_AFFINE_RELU * 100 + 8:      return dispatch_nb<1, 8, LD_FMN_INPUT, EP_AFFINE_RELU>(a, nb, st);
    // the first layer factorised: its pre-activation maps (fp32, one or two pseudo-planes) and the two consumers that synthesise its output
    case 10000 + LD_FMN_INPUT * 1000 + EP_AFFINE_F32_NHWC * 100 + 8:  return nb == 1 ? launch<1, 8, LD_FMN_INPUT, EP_AFFINE_F32_NHWC, 1, 8, 32>(a, st) : MPF_ERR_UNSUPPORTED;
    case 20000 + LD_FMN_SYNTH * 1000 + EP_AFFINE_RELU * 100 + 16:     return nb == 2 ? launch<2, 16, LD_FMN_SYNTH, EP_AFFINE_RELU, 2, 4, 32>(a, st) : MPF_ERR_UNSUPPORTED;
    case 10000 + LD_BILINEAR_SYNTH * 1000 + EP_AFFINE_RELU * 100 + 16: return nb == 1 ? launch<1, 16, LD_BILINEAR_SYNTH, EP_AFFINE_RELU, 1, 8, 32>(a, st) : MPF_ERR_UNSUPPORTED;
    case 20000 + LD_DIRECT * 1000 + EP_AFFINE_RELU * 100 + 16:        return dispatch_nb<2, 16, LD_DIRECT, EP_AFFINE_RELU>(a, nb, st);
    case 20000 + LD_DIRECT * 1000 + EP_AFFINE_RELU * 100 + 32:        return dispatch_nb<2, 32, LD_DIRECT, EP_AFFINE_RELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_AFFINE_RELU * 100 + 32:        return dispatch_nb<1, 32, LD_DIRECT, EP_AFFINE_RELU>(a, nb, st);
    case 10000 + LD_BILINEAR_CAT * 1000 + EP_AFFINE_RELU * 100 + 32:  return dispatch_nb<1, 32, LD_BILINEAR_CAT, EP_AFFINE_RELU>(a, nb, st);
    case 10000 + LD_BILINEAR_CAT * 1000 + EP_AFFINE_RELU * 100 + 16:  return dispatch_nb<1, 16, LD_BILINEAR_CAT, EP_AFFINE_RELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_AFFINE_RELU_F32 * 100 + 16:    return dispatch_nb<1, 16, LD_DIRECT, EP_AFFINE_RELU_F32>(a, nb, st);
    // gated decoder (reflection padding)
    case 10000 + LD_NEAREST_PLANE * 1000 + EP_GATED_ELU * 100 + 32:   return dispatch_nb<1, 32, LD_NEAREST_PLANE, EP_GATED_ELU>(a, nb, st);
    case 10000 + LD_NEAREST_PLANE * 1000 + EP_GATED_ELU * 100 + 16:   return dispatch_nb<1, 16, LD_NEAREST_PLANE, EP_GATED_ELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_GATED_ELU * 100 + 32:          return dispatch_nb<1, 32, LD_DIRECT, EP_GATED_ELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_GATED_ELU * 100 + 16:          return dispatch_nb<1, 16, LD_DIRECT, EP_GATED_ELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_AFFINE_RELU * 100 + 16:        return dispatch_nb<1, 16, LD_DIRECT, EP_AFFINE_RELU>(a, nb, st);
    case 10000 + LD_DIRECT * 1000 + EP_GATED_PLANAR_F32 * 100 + 16:   return dispatch_nb<1, 16, LD_DIRECT, EP_GATED_PLANAR_F32>(a, nb, st);
    case 10000 + LD_NEAREST_PLANE * 1000 + EP_GATED_ELU_PAIRED * 100 + 16:
        return nb == 3 ? launch<1, 16, LD_NEAREST_PLANE, EP_GATED_ELU_PAIRED, 3, 8, 32>(a, st) : MPF_ERR_UNSUPPORTED;
    case 10000 + LD_DIRECT * 1000 + EP_GATED_ELU_PAIRED * 100 + 16:
        return nb == 3 ? launch<1, 16, LD_DIRECT, EP_GATED_ELU_PAIRED, 3, 8, 32>(a, st) : MPF_ERR_UNSUPPORTED;
    case 10000 + LD_DIRECT * 1000 + EP_GATED_PLANAR_F32_PAIRED * 100 + 16:
        return nb == 1 && a.Cst <= 8 ? launch<1, 16, LD_DIRECT, EP_GATED_PLANAR_F32_PAIRED, 1, 8, 32>(a, st) : MPF_ERR_UNSUPPORTED;
    }
    mpf_set_error("mpf_conv3x3_f16: combination loader=%d epilogue=%d ct=%d stride=%d is not built", a.loader, a.epi, a.ct, a.stride);
    return MPF_ERR_UNSUPPORTED;
}
