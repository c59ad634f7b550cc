// mpf_encoder.hip - the single-image part of the MPI producer network in fp32: the RGBD ResNet-18 encoder
// (reference model/CPN/encoder.py:20-101) and the decoder's bottleneck (model/CPN/decoder.py:85-88,131-138).
//
// These layers see ONE image (not S plane-images) and 1/2 .. 1/128 of its resolution: 18 GMAC spread over 24 convolutions whose
// widest GEMM is 512 x 480 x 4608.  They feed the gated decoder through a random-access skip path, and an fp16 encoder costs 3x the
// end-to-end error of the producer (mpiflow_amd/model/engine.py, HipPredictor), so the arithmetic here is fp32 throughout:
// v_mfma_f32_16x16x4_f32, fp32 NHWC activations, BatchNorm folded into a per-channel scale / shift in the epilogue.
//
// k_conv2d_f32: implicit GEMM, one workgroup = 32 output channels x 32 output pixels, its four waves SPLIT K (the taps x input
// channels) between them and reduce through LDS - the deep layers have 480 .. 1920 pixels, i.e. 240 .. 480 tiles for 256 CUs, and K up
// to 4608, so splitting K is what fills the chip.  No input tile in LDS: a wave reads its operands as 16-byte vectors straight from
// L1 / L2 (the whole encoder's activations and weights are 70 MB, resident in the 256 MB MALL / 4 MB L2s after the first touch):
// lane (m = l % 16, g = l / 16) loads 4 consecutive input channels of pixel m at the tap of K-vector 4 step + g, and the matching 4
// weights of output channel m; the 4 elements feed 4 MFMAs (any permutation of K is a valid GEMM as long as both operands use it).
#include "mpf_common.h"
#include <hip/hip_fp16.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_encoder_input(const float *__restrict__ image, const float *__restrict__ disp, int N, f32x4 *__restrict__ out)
{
    // model/CPN/encoder.py:89-93: x = cat((image - mean) / std, disparity); the ImageNet constants of :84-85
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    f32x4 v;
    v.x = (image[n] - 0.485f) / 0.229f;
    v.y = (image[N + n] - 0.456f) / 0.224f;
    v.z = (image[2 * N + n] - 0.406f) / 0.225f;
    v.w = disp[n];
    out[n] = v;
}

__device__ __forceinline__ float act_apply(float y, int act, float slope)
{
    if (act == 1) return fmaxf(y, 0.f);
    if (act == 2) return y > 0.f ? y : y * slope;
    return y;
}

template <int KS, int NW, int U>
__global__ __launch_bounds__(NW * 64) void k_conv2d_f32(const MpfConv2dArgs a)
{
    __shared__ __attribute__((aligned(16))) float part[NW][32][36];  // [wave][pixel][channel], pitch 36: 16-byte rows, spread over the banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
    const int P = a.Hout * a.Wout, p0 = blockIdx.x * 32, cb = blockIdx.y;          // cb: 32-channel output block
    const int V = a.Cin >> 2, lgV = 31 - __clz(V), nsteps = (KS * KS * V + 3) >> 2;
    const int Ws = a.Win >> a.up;
    int oy[2], ox[2];
    bool pv[2];
#pragma unroll
    for (int pg = 0; pg < 2; ++pg) {
        const int p = p0 + 16 * pg + m;
        pv[pg] = p < P;
        const int pc = pv[pg] ? p : 0;
        oy[pg] = pc / a.Wout;
        ox[pg] = pc - oy[pg] * a.Wout;
        oy[pg] = oy[pg] * a.stride - a.pad;
        ox[pg] = ox[pg] * a.stride - a.pad;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) acc[nb][pg] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.wpack) + (size_t)(cb * 2) * nsteps * 64 + lane;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(a.src);
    // U steps per trip: all 4 U operand loads are issued before the first MFMA of the trip, so that a wave with few co-resident waves (the deep
    // layers: one or two workgroups per CU) does not pay an L2 round trip per step; the shallow layers (thousands of workgroups) hide that
    // round trip by occupancy and are faster with the registers left free (U = 1)
    for (int s0 = wave; s0 < nsteps; s0 += NW * U) {
        f32x4 xv[U][2], wv[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u * NW;
            const bool live = s < nsteps;                           // past the end: zero pixels against the last step's (finite) weights
            const int sc = live ? s : nsteps - 1;
            const int v = 4 * sc + g, tap = v >> lgV, c4 = v & (V - 1);
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                const int iy = oy[pg] + ky, ix = ox[pg] + kx;
                const bool ok = live && pv[pg] && tap < KS * KS && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                const int sy = ok ? iy >> a.up : 0, sx = ok ? ix >> a.up : 0;
                const f32x4 ld = src[(size_t)(sy * Ws + sx) * V + (ok ? c4 : 0)];
                xv[u][pg] = ok ? ld : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wv[u][nb] = wp[(size_t)(nb * nsteps + sc) * 64];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int pg = 0; pg < 2; ++pg) acc[nb][pg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][nb][j], xv[u][pg][j], acc[nb][pg], 0, 0, 0);
    }
    // lane (m, g) of accumulator (nb, pg) holds output channels 16 nb + 4 g .. + 3 of pixel 16 pg + m
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) *reinterpret_cast<f32x4 *>(&part[wave][16 * pg + m][16 * nb + 4 * g]) = acc[nb][pg];
    __syncthreads();
    const int pix = tid >> 3, c = (tid & 7) * 4, p = p0 + pix;
    if (tid >= 256 || p >= P) return;
    f32x4 y = *reinterpret_cast<const f32x4 *>(&part[0][pix][c]);
#pragma unroll
    for (int w = 1; w < NW; ++w) {                                  // fixed order: the sum does not depend on scheduling
        const f32x4 t = *reinterpret_cast<const f32x4 *>(&part[w][pix][c]);
        y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
    }
    const int ch = cb * 32 + c;
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.scale + ch), sh = *reinterpret_cast<const f32x4 *>(a.shift + ch);
    y.x = fmaf(y.x, sc.x, sh.x); y.y = fmaf(y.y, sc.y, sh.y); y.z = fmaf(y.z, sc.z, sh.z); y.w = fmaf(y.w, sc.w, sh.w);
    const size_t o = (size_t)p * a.Cout + ch;
    if (a.residual) {
        const f32x4 r = *reinterpret_cast<const f32x4 *>(a.residual + o);
        y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
    }
    y.x = act_apply(y.x, a.act, a.slope); y.y = act_apply(y.y, a.act, a.slope);
    y.z = act_apply(y.z, a.act, a.slope); y.w = act_apply(y.w, a.act, a.slope);
    if (a.out) *reinterpret_cast<f32x4 *>(a.out + o) = y;
    if (a.out_f16) {
        const __half2 lo = __floats2half2_rn(y.x, y.y), hi = __floats2half2_rn(y.z, y.w);
        uint2 pk;
        pk.x = *reinterpret_cast<const unsigned *>(&lo);
        pk.y = *reinterpret_cast<const unsigned *>(&hi);
        *reinterpret_cast<uint2 *>(reinterpret_cast<__half *>(a.out_f16) + o) = pk;
    }
}

__global__ __launch_bounds__(256) void k_maxpool3x3s2_f32(const f32x4 *__restrict__ src, int Hin, int Win, int V, int Hout, int Wout, f32x4 *__restrict__ out)
{
    // nn.MaxPool2d(3, 2, 1) (model/CPN/encoder.py:97 via torchvision's ResNet stem; model/CPN/decoder.py:81): padding never wins the max
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= Hout * Wout * V) return;
    const int c4 = n % V, p = n / V, oy = p / Wout, ox = p - oy * Wout;
    f32x4 best = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy - 1 + ky;
        if (iy < 0 || iy >= Hin) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * ox - 1 + kx;
            if (ix < 0 || ix >= Win) continue;
            const f32x4 t = src[(size_t)(iy * Win + ix) * V + c4];
            best.x = fmaxf(best.x, t.x); best.y = fmaxf(best.y, t.y); best.z = fmaxf(best.z, t.z); best.w = fmaxf(best.w, t.w);
        }
    }
    out[n] = best;
}

}  // namespace

extern "C" int mpf_encoder_input(const float *d_image_3HW, const float *d_disp_HW, int H, int W, float *d_out_HW4, void *stream)
{
    MPF_REQUIRE(d_image_3HW && d_disp_HW && d_out_HW4 && H >= 1 && W >= 1, "mpf_encoder_input: bad argument");
    MPF_REQUIRE(mpf_aligned16(d_out_HW4), "mpf_encoder_input: output must be 16-byte aligned");
    const int N = H * W;
    hipLaunchKernelGGL(k_encoder_input, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_image_3HW, d_disp_HW, N, reinterpret_cast<f32x4 *>(d_out_HW4));
    return mpf_launch_status("k_encoder_input");
}

extern "C" int mpf_conv2d_f32(const MpfConv2dArgs *args, void *stream)
{
    MPF_REQUIRE(args != nullptr, "mpf_conv2d_f32: null argument block");
    const MpfConv2dArgs &a = *args;
    MPF_REQUIRE(a.src && a.wpack && a.scale && a.shift && (a.out || a.out_f16), "mpf_conv2d_f32: null source / weights / epilogue rows / output");
    MPF_REQUIRE(a.ksize == 1 || a.ksize == 3 || a.ksize == 7, "mpf_conv2d_f32: kernel size must be 1, 3 or 7");
    MPF_REQUIRE((a.stride == 1 || a.stride == 2) && a.pad >= 0 && a.pad <= a.ksize / 2 && (a.up == 0 || a.up == 1), "mpf_conv2d_f32: bad stride / padding / upsampling");
    MPF_REQUIRE(a.Cin >= 4 && (a.Cin & (a.Cin - 1)) == 0, "mpf_conv2d_f32: input channels must be a power of two >= 4");
    MPF_REQUIRE(a.Cout >= 32 && a.Cout % 32 == 0 && a.Cout / 32 <= 65535, "mpf_conv2d_f32: output channels must be a multiple of 32");
    MPF_REQUIRE(a.Hin >= 1 && a.Win >= 1 && (a.up == 0 || (a.Hin % 2 == 0 && a.Win % 2 == 0)), "mpf_conv2d_f32: bad input size");
    MPF_REQUIRE(a.Hout == (a.Hin + 2 * a.pad - a.ksize) / a.stride + 1 && a.Wout == (a.Win + 2 * a.pad - a.ksize) / a.stride + 1 && a.Hout >= 1 && a.Wout >= 1,
                "mpf_conv2d_f32: output size does not match the convolution");
    MPF_REQUIRE((size_t)a.Hin * a.Win * a.Cin < 0x7FFFFFFFull && (size_t)a.Hout * a.Wout * a.Cout < 0x7FFFFFFFull, "mpf_conv2d_f32: tensor too large");
    MPF_REQUIRE(a.act >= 0 && a.act <= 2, "mpf_conv2d_f32: activation must be 0 (none), 1 (ReLU) or 2 (leaky ReLU)");
    MPF_REQUIRE(mpf_aligned16(a.src) && mpf_aligned16(a.wpack) && mpf_aligned16(a.scale) && mpf_aligned16(a.shift) && mpf_aligned16(a.residual) && mpf_aligned16(a.out) &&
                    (((uintptr_t)a.out_f16) & 7) == 0, "mpf_conv2d_f32: buffers must be 16-byte aligned");
    const dim3 grid((a.Hout * a.Wout + 31) / 32, a.Cout / 32);
    hipStream_t st = (hipStream_t)stream;
    // few tiles and a long K (the 1/32 .. 1/128 layers): 8 waves split K instead of 4, four steps' loads in flight per wave (the
    // result depends on the variant chosen, which is a function of the shapes only)
    const int nsteps = (a.ksize * a.ksize * (a.Cin / 4) + 3) / 4;
    const bool wide = (int)(grid.x * grid.y) <= 512 && nsteps >= 128;
    if (a.ksize == 1) {
        if (wide) hipLaunchKernelGGL((k_conv2d_f32<1, 8, 4>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_conv2d_f32<1, 4, 1>), grid, dim3(256), 0, st, a);
    } else if (a.ksize == 3) {
        if (wide) hipLaunchKernelGGL((k_conv2d_f32<3, 8, 4>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_conv2d_f32<3, 4, 1>), grid, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_conv2d_f32<7, 4, 1>), grid, dim3(256), 0, st, a);
    }
    return mpf_launch_status("k_conv2d_f32");
}

extern "C" int mpf_maxpool3x3s2_f32(const float *d_src_HWC, int Hin, int Win, int C, float *d_out, void *stream)
{
    MPF_REQUIRE(d_src_HWC && d_out && Hin >= 1 && Win >= 1 && C >= 4 && C % 4 == 0, "mpf_maxpool3x3s2_f32: bad argument");
    MPF_REQUIRE(mpf_aligned16(d_src_HWC) && mpf_aligned16(d_out), "mpf_maxpool3x3s2_f32: buffers must be 16-byte aligned");
    const int Hout = (Hin - 1) / 2 + 1, Wout = (Win - 1) / 2 + 1, V = C / 4;
    MPF_REQUIRE((size_t)Hin * Win * C < 0x7FFFFFFFull, "mpf_maxpool3x3s2_f32: tensor too large");
    const int n = Hout * Wout * V;
    hipLaunchKernelGGL(k_maxpool3x3s2_f32, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4 *>(d_src_HWC), Hin, Win, V, Hout, Wout,
                       reinterpret_cast<f32x4 *>(d_out));
    return mpf_launch_status("k_maxpool3x3s2_f32");
}
