// mpf_render.hip - fused MPI render / flow kernels for MI355X (gfx950, wave64) and their C-ABI launchers.
//
//   k_src_blend_flow   Stage A + C  (source frame: transmittance chain -> blended RGBA stack + volume-rendered flows)
//   k_warp_composite   Stage B      (target frame: per-plane homography warp + front-to-back composite)
//   k_mask_quads, k_merge, k_to_u8_bgr   small per-pixel helpers around them
//
// All of them are HBM-streaming kernels (30-40 flop per 16 B); none is GEMM-shaped, so no MFMA.  What matters is
// coalesced 16-byte accesses, enough waves in flight to cover gather latency, an XCD-aware tile order so the texel
// rows two neighbouring tiles share are found in the same L2, and keeping the whole S-plane recurrence in
// registers so that no [S,...] intermediate is ever written (the reference materialises ~8 of them per view).
#include <string.h>
#include <type_traits>
#include "mpf_common.h"
#include "mpf_math.h"

// PRODUCT vs WITNESS build.  libmpiflow_hip.so (the product) is built WITHOUT MPF_WITNESS: every bench-only knob is a compile-time constant there, mpf_tune refuses the
// keys that select retired kernel variants or timing ablations, and those kernels are not compiled.  libmpiflow_hip_witness.so (-DMPF_WITNESS, loaded by the tests and
// tools that A/B the retired forms against the shipped ones) keeps them all.  MPF_WIT(x): a run-time knob in the witness build, the constant 0 in the product.
#ifdef MPF_WITNESS
#define MPF_WIT(x) (x)
#define MPF_KNOB static int
#else
#define MPF_WIT(x) 0
#define MPF_KNOB static constexpr int
#endif

#define MPF_TILE_W 64   // one wavefront = 64 consecutive target pixels of one row
#define MPF_TILE_H 4    // 4 waves per 256-thread workgroup

// ---------------------------------------------------------------------------------------------------------------
// Stage B
// ---------------------------------------------------------------------------------------------------------------

struct MpfPlaneTaps {
    int o00, o01, o10, o11;   // texel offsets (y*W + x) of the four taps, clamped in range
    bool e_in, s_in, valid;
    float nw, ne, sw, se;
    float X, Y, Z;            // xyz_tgt at the clamped source coordinate
};

// Coordinates of target pixel (fx,fy) in source plane s, bilinear taps, validity, and the warped xyz_tgt.
// reference: utils/mpi/homography_sampler.py:131-147 (H_src_tgt . p, divide, valid), :151-156 (normalise +
// grid_sample), and the three xyz channels of mpi_rendering.py:288-301 evaluated analytically:
// xyz_tgt = G . (K^-1 (ix,iy,1) * d_s ; 1)  (mpi_rendering.py:213-256 at the clamped coordinate).
MPF_DEV MpfPlaneTaps mpf_plane_taps(const float *__restrict__ params, int s, float fx, float fy, int W, int H)
{
    const float *rec = params + MPF_PARAMS_HEADER + MPF_PLANE_RECORD * s;
    float qx = mpf_row3_xy1(rec[0], rec[1], rec[2], fx, fy);
    float qy = mpf_row3_xy1(rec[3], rec[4], rec[5], fx, fy);
    float qz = mpf_row3_xy1(rec[6], rec[7], rec[8], fx, fy);
    float u = qx / qz, v = qy / qz;
    MpfPlaneTaps p;
    p.valid = (u < (float)W) && (u > -1.0f) && (v < (float)H) && (v > -1.0f);
    MpfTaps t = mpf_make_taps(u, v, W, H);
    p.e_in = t.e_in; p.s_in = t.s_in;
    p.nw = t.nw; p.ne = t.ne; p.sw = t.sw; p.se = t.se;
    p.o00 = t.y0 * W + t.x0;
    p.o01 = p.o00 + (t.e_in ? 1 : 0);
    p.o10 = p.o00 + (t.s_in ? W : 0);
    p.o11 = p.o10 + (t.e_in ? 1 : 0);
    const float d = rec[9];
    float rx = mpf_row3_xy1(params[0], params[1], params[2], t.ix, t.iy) * d;
    float ry = mpf_row3_xy1(params[3], params[4], params[5], t.ix, t.iy) * d;
    float rz = mpf_row3_xy1(params[6], params[7], params[8], t.ix, t.iy) * d;
    p.X = mpf_row4_xyz1(params[9], params[10], params[11], params[12], rx, ry, rz);
    p.Y = mpf_row4_xyz1(params[13], params[14], params[15], params[16], rx, ry, rz);
    p.Z = mpf_row4_xyz1(params[17], params[18], params[19], params[20], rx, ry, rz);
    return p;
}

struct MpfRaw {          // the 4 taps x 4 channels of one plane as fetched, plus the mask quad
    float4 t00, t01, t10, t11;
    float4 mq;
};

template <bool INTERLEAVED, bool HAS_MASK>
MPF_DEV MpfRaw mpf_fetch(const float *__restrict__ rgba, const float *__restrict__ quads, int s, int64_t N,
                         const MpfPlaneTaps &p)
{
    MpfRaw r;
    if (INTERLEAVED) {
        const float4 *pl = reinterpret_cast<const float4 *>(rgba) + (int64_t)s * N;
        r.t00 = pl[p.o00]; r.t01 = pl[p.o01]; r.t10 = pl[p.o10]; r.t11 = pl[p.o11];
    } else {
        const float *pl = rgba + (int64_t)s * 4 * N;
        r.t00 = make_float4(pl[p.o00], pl[N + p.o00], pl[2 * N + p.o00], pl[3 * N + p.o00]);
        r.t01 = make_float4(pl[p.o01], pl[N + p.o01], pl[2 * N + p.o01], pl[3 * N + p.o01]);
        r.t10 = make_float4(pl[p.o10], pl[N + p.o10], pl[2 * N + p.o10], pl[3 * N + p.o10]);
        r.t11 = make_float4(pl[p.o11], pl[N + p.o11], pl[2 * N + p.o11], pl[3 * N + p.o11]);
    }
    if (HAS_MASK) r.mq = reinterpret_cast<const float4 *>(quads)[p.o00];
    return r;
}

MPF_DEV float mpf_tap4(const MpfPlaneTaps &p, float a, float b, float c, float d)
{
    // out-of-range neighbours are read as 0 (ATen masks them), their weight is 0 as well
    float o = a * p.nw;
    o = fmaf(p.e_in ? b : 0.0f, p.ne, o);
    o = fmaf(p.s_in ? c : 0.0f, p.sw, o);
    o = fmaf((p.e_in && p.s_in) ? d : 0.0f, p.se, o);
    return o;
}

template <bool INTERLEAVED, bool HAS_MASK, int NL>
__global__ void __launch_bounds__(MPF_TILE_W *MPF_TILE_H)
k_warp_composite(const float *__restrict__ rgba, const float *__restrict__ quads, const float *__restrict__ params,
                 int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                 float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out)
{
    const int64_t N = (int64_t)H * W;
    const unsigned tiles_x = (W + MPF_TILE_W - 1) / MPF_TILE_W;
    const unsigned tile = mpf_xcd_remap(blockIdx.x, gridDim.x);
    const int x = (tile % tiles_x) * MPF_TILE_W + (threadIdx.x & (MPF_TILE_W - 1));
    const int y = (tile / tiles_x) * MPF_TILE_H + (threadIdx.x / MPF_TILE_W);
    const bool active = (x < W) && (y < H);
    // threads past the image edge shadow the last pixel (in-range loads, no store)
    const float fx = (float)min(x, W - 1), fy = (float)min(y, H - 1);

    MpfPlaneTaps cur = mpf_plane_taps(params, 0, fx, fy, W, H);
    MpfRaw raw = mpf_fetch<INTERLEAVED, HAS_MASK>(rgba, quads, 0, N, cur);

    double acc = 1.0;                       // torch.cumprod keeps its running product in double on CPU
    MpfCsum<NL> cw, cd, co, c0, c1, c2;
    cw.init(); cd.init(); co.init(); c0.init(); c1.init(); c2.init();
    float nvalid = 0.0f;

    for (int s = 0; s < S; ++s) {
        // 1. geometry of plane s+1 (pure ALU) and its gathers, issued before plane s is composited
        MpfPlaneTaps nxt = cur;
        MpfRaw raw_n = raw;
        float dist = 1e3f;                                         // mpi_rendering.py:73-78
        if (s + 1 < S) {
            nxt = mpf_plane_taps(params, s + 1, fx, fy, W, H);
            raw_n = mpf_fetch<INTERLEAVED, HAS_MASK>(rgba, quads, s + 1, N, nxt);
            dist = mpf_norm3(nxt.X - cur.X, nxt.Y - cur.Y, nxt.Z - cur.Z);   // :68-70
        }
        // 2. plane s: bilinear taps -> warped rgb, sigma, mask
        float r = mpf_tap4(cur, raw.t00.x, raw.t01.x, raw.t10.x, raw.t11.x);
        float g = mpf_tap4(cur, raw.t00.y, raw.t01.y, raw.t10.y, raw.t11.y);
        float b = mpf_tap4(cur, raw.t00.z, raw.t01.z, raw.t10.z, raw.t11.z);
        float sg = mpf_tap4(cur, raw.t00.w, raw.t01.w, raw.t10.w, raw.t11.w);
        sg = (cur.Z >= 0.0f) ? sg : 0.0f;                          // :336-338
        // 3. front-to-back composite step                          // :79-90
        float T = mpf_expf(-sg * dist);
        float alpha = 1.0f - T;
        float tacc = (float)acc;
        float w = tacc * alpha;
        acc *= (double)(T + 1e-6f);
        cw.push(w);
        c0.push(w * r); c1.push(w * g); c2.push(w * b);
        cd.push(w * cur.Z);
        if (HAS_MASK) {
            float om = raw.mq.x * cur.nw;        // quads carry the zeros of out-of-range neighbours already
            om = fmaf(raw.mq.y, cur.ne, om);
            om = fmaf(raw.mq.z, cur.sw, om);
            om = fmaf(raw.mq.w, cur.se, om);
            co.push(w * om);
        }
        nvalid += cur.valid ? 1.0f : 0.0f;                         // :347
        if (((s + 1) & 15) == 0) {
            cw.fold(s + 1); cd.fold(s + 1); c0.fold(s + 1); c1.fold(s + 1); c2.fold(s + 1);
            if (HAS_MASK) co.fold(s + 1);
        }
        cur = nxt;
        raw = raw_n;
    }
    if (active) {
        const int64_t n = (int64_t)y * W + x;
        rgb_out[n] = c0.final();
        rgb_out[N + n] = c1.final();
        rgb_out[2 * N + n] = c2.final();
        if (u8_out) { u8_out[3 * n] = mpf_to_u8(c2.final()); u8_out[3 * n + 1] = mpf_to_u8(c1.final()); u8_out[3 * n + 2] = mpf_to_u8(c0.final()); }
        if (depth_out) depth_out[n] = cd.final() / (cw.final() + 1e-5f);   // :152
        if (HAS_MASK) om_out[n] = co.final();
        if (tgt_mask_out) tgt_mask_out[n] = nvalid;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage B, version 2 (interleaved RGBA stack only): same arithmetic as k_warp_composite, restructured for the machine
//   * two register sets (A/B) ping-pong across an x2-unrolled plane loop: the gathers of plane s+1 are issued a whole
//     composite step before their first use, and nothing is copied (v1's `cur = nxt` forced hipcc to wait for the loads
//     right after issuing them)
//   * the four divisions per plane share refined reciprocals and skip the denormal/overflow guards (mpf_div_nr), sqrt and
//     exp likewise (mpf_sqrt_nr, mpf_expf_fast) - bit-identical results in the range the path operates in
//   * out-of-range bilinear neighbours: their weight is exactly 0 (x0+1 == W only when ix == W-1 exactly), the tap
//     re-reads the in-range texel and 0 * finite adds nothing - no selects (rgb, sigma >= 0, finite)
//   * plane base pointers stay scalar (SGPR base + 32-bit VGPR byte offset), validity is counted where it is computed
// ---------------------------------------------------------------------------------------------------------------

// d_params is read-only for the whole launch.  hipcc proves that for a plain global pointer only while nothing in the kernel could
// alias it: behind a workgroup barrier (the fence of __syncthreads() counts as a clobber), or inside a kernel that also carries another
// role with stores through unrelated pointers (k_pair_overlap), the per-plane records turn into VECTOR loads issued right before their
// use - measured: 3-6 extra VMEM instructions and a full memory latency per wave and plane, plus the registers to hold them (spills at
// 5 waves per SIMD).  Reading them through the constant address space keeps them on the scalar unit, whatever surrounds the body.
typedef const __attribute__((address_space(4))) float *MpfConstParams;

struct MpfGeom {
    float nw, ne, sw, se;
    float X, Y, Z;
    unsigned b00, b01, b10, b11;     // byte offsets of the four taps inside one [H,W,4] fp32 plane
};

#ifndef MPF_XYZ_MFMA
#define MPF_XYZ_MFMA 0          // 1: xyz_tgt = G . (r; 1) on the matrix cores (4 x v_mfma_f32_4x4x1 instead of 12 VALU ops), see mpf_geom_core
#endif
struct MpfConsts {
    float fx, fy;
    float halfW, halfH, rhalfW, rhalfH, maxx, maxy;
    float Wf, Hf;
    int W, H;
    unsigned row_bytes;
#if MPF_XYZ_MFMA
    float G0, G1, G2, G3;        // column k of G_tgt_src's 3x4 block, row (lane % 4) - the A operands of the 4x4x1 outer products (row 3: zero)
#endif
};

template <class ParamPtr>
MPF_DEV void mpf_consts_pose(MpfConsts &c, const ParamPtr params)
{
#if MPF_XYZ_MFMA
    const unsigned r = threadIdx.x & 3u;
    c.G0 = r == 0 ? params[9] : (r == 1 ? params[13] : (r == 2 ? params[17] : 0.0f));
    c.G1 = r == 0 ? params[10] : (r == 1 ? params[14] : (r == 2 ? params[18] : 0.0f));
    c.G2 = r == 0 ? params[11] : (r == 1 ? params[15] : (r == 2 ? params[19] : 0.0f));
    c.G3 = r == 0 ? params[12] : (r == 1 ? params[16] : (r == 2 ? params[20] : 0.0f));
#endif
}

// KS: K_src^-1 has the pinhole form [[a,0,b],[0,c,e],[0,0,1]] (checked at run time, uniform).  Then the reference's dense
//     3x3 . (ix,iy,1) chain collapses exactly: a*ix + 0*iy is a*ix, 0*ix + c*iy is c*iy, the third row is exactly 1, and
//     1 * d is d - same bits, 6 fewer VALU ops per plane.
// TP: the stack is followed by >= (W+1) texels of finite padding, so the east / south taps can always be read at
//     +16 / +row bytes (immediate offsets); where they fall outside the image their weight is exactly 0.
// The arithmetic of one plane for one target pixel: source coordinate, bilinear weights, north-west texel (x0, y0) and the
// warped xyz_tgt at the clamped coordinate.  Shared by the gather kernels and the LDS-staged kernel (and by the latter's
// tile-corner evaluation), so every variant performs the identical IEEE op sequence.
template <bool KS, bool WANT_VALID, class ParamPtr>
MPF_DEV float mpf_geom_core(const ParamPtr params, int s, const MpfConsts &c, float &nw, float &ne, float &sw, float &se,
                            float &X, float &Y, float &Z, int &x0, int &y0, float &qz_out)
{
    const ParamPtr rec = params + MPF_PARAMS_HEADER + MPF_PLANE_RECORD * s;
    float qx = mpf_row3_xy1(rec[0], rec[1], rec[2], c.fx, c.fy);
    float qy = mpf_row3_xy1(rec[3], rec[4], rec[5], c.fx, c.fy);
    float qz = mpf_row3_xy1(rec[6], rec[7], rec[8], c.fx, c.fy);
    qz_out = qz;
    const float rz = mpf_rcp_nr(qz);
    float u = mpf_div_nr(qx, qz, rz), v = mpf_div_nr(qy, qz, rz);
    float valid = 0.0f;
    if (WANT_VALID) {
        const bool inside = (u < c.Wf) & (u > -1.0f) & (v < c.Hf) & (v > -1.0f);
        valid = inside ? 1.0f : 0.0f;
    }
    float gx = mpf_div_by_const(u + 0.5f, c.halfW, c.rhalfW) - 1.0f;
    float gy = mpf_div_by_const(v + 0.5f, c.halfH, c.rhalfH) - 1.0f;
    float ix = (gx + 1.0f) * c.halfW - 0.5f;
    float iy = (gy + 1.0f) * c.halfH - 0.5f;
    ix = __builtin_amdgcn_fmed3f(ix, 0.0f, c.maxx);       // min(max_val, max(x, 0)) in one op (inputs are never NaN here)
    iy = __builtin_amdgcn_fmed3f(iy, 0.0f, c.maxy);
    // ix >= 0: x - floor(x) is exact and equals v_fract_f32(x); trunc == floor
    float w = __builtin_amdgcn_fractf(ix), e = 1.0f - w;
    float n = __builtin_amdgcn_fractf(iy), sgt = 1.0f - n;
    nw = sgt * e; ne = sgt * w; sw = n * e; se = n * w;
    x0 = (int)ix; y0 = (int)iy;
    const float d = rec[9];
    float rx, ry, rzz;
    if (KS) {
        rx = (params[0] * ix + params[2]) * d;
        ry = (params[4] * iy + params[5]) * d;
        rzz = d;
    } else {
        rx = mpf_row3_xy1(params[0], params[1], params[2], ix, iy) * d;
        ry = mpf_row3_xy1(params[3], params[4], params[5], ix, iy) * d;
        rzz = mpf_row3_xy1(params[6], params[7], params[8], ix, iy) * d;
    }
#if MPF_XYZ_MFMA
    // v_mfma_f32_4x4x1_16b_f32: D_v(lane) = A(lane 4 * (lane / 4) + v) * B(lane) + C_v(lane), one exact IEEE fma per element - identical to
    // v_fma_f32 on 2 * 10^10 operand triples of this path's ranges (tools/mfma_fma_exact.hip, profiles/r3/mfma_fma_exact.log) - so the chain
    // a0*X, fma(a1,Y,.), fma(a2,Z,.), fma(a3,1,.) of the three rows of G becomes 4 dependent outer products: rows in the accumulator index,
    // this lane's (rx, ry, rz, 1) as B.  (The first step is fma(a0, X, +0): equal to the product except for the sign of an exact zero.)
    typedef float mpf_f4 __attribute__((ext_vector_type(4)));
    mpf_f4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(c.G0, rx, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(c.G1, ry, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(c.G2, rzz, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(c.G3, 1.0f, acc, 0, 0, 0);
    X = acc[0]; Y = acc[1]; Z = acc[2];
#else
    X = mpf_row4_xyz1(params[9], params[10], params[11], params[12], rx, ry, rzz);
    Y = mpf_row4_xyz1(params[13], params[14], params[15], params[16], rx, ry, rzz);
    Z = mpf_row4_xyz1(params[17], params[18], params[19], params[20], rx, ry, rzz);
#endif
    return valid;
}

template <bool KS, bool TP, bool WANT_VALID = true, bool PLANAR = false>
MPF_DEV float mpf_geom(const MpfConstParams params, int s, const MpfConsts &c, MpfGeom &g)
{
    int x0, y0;
    float qz;
    const float valid = mpf_geom_core<KS, WANT_VALID>(params, s, c, g.nw, g.ne, g.sw, g.se, g.X, g.Y, g.Z, x0, y0, qz);
    const unsigned o00 = __umul24((unsigned)y0, (unsigned)c.W) + (unsigned)x0;   // H*W < 2^27, W < 2^24
    if (PLANAR) {                        // [S,4,H,W] stack: b00 = byte offset inside one fp32 channel plane, b01 = inside the [H,W,4] mask quads
        g.b00 = o00 * 4u;
        g.b01 = o00 * 16u;
        return valid;
    }
    g.b00 = o00 * 16u;
    if (TP) {
        g.b01 = g.b00 + 16u;
        g.b10 = g.b00 + c.row_bytes;
        g.b11 = g.b10 + 16u;
    } else {
        const unsigned dx = (x0 + 1 < c.W) ? 16u : 0u;
        const unsigned dy = (y0 + 1 < c.H) ? c.row_bytes : 0u;
        g.b01 = g.b00 + dx;
        g.b10 = g.b00 + dy;
        g.b11 = g.b10 + dx;
    }
    return valid;
}

struct MpfRaw4 { float4 t00, t01, t10, t11, mq; };

typedef unsigned mpf_v4u __attribute__((ext_vector_type(4)));

// TP (tail-padded stack): the four taps are ONE address computation - raw buffer loads take the tap's +16 as the instruction's
// immediate offset and the +row as its scalar offset, where flat/global loads would need three more 32-bit VALU adds per plane
// (the compiler cannot fold them: a wrapped 32-bit offset would address differently).
template <bool HAS_MASK, bool TP>
MPF_DEV void mpf_fetch2(const char *__restrict__ plane, const char *__restrict__ quads, const MpfGeom &g, MpfRaw4 &r, const unsigned row_bytes,
                        const unsigned span)
{
    if (TP) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(plane), 0, span, 0x00020000);
        r.t00 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, g.b00, 0, 0));
        r.t01 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, g.b00 + 16u, 0, 0));
        r.t10 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, g.b00, row_bytes, 0));
        r.t11 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, g.b00 + 16u, row_bytes, 0));
        if (HAS_MASK) {
            __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(quads), 0, span, 0x00020000);
            r.mq = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rq, g.b00, 0, 0));
        }
    } else {
        r.t00 = *reinterpret_cast<const float4 *>(plane + g.b00);
        r.t01 = *reinterpret_cast<const float4 *>(plane + g.b01);
        r.t10 = *reinterpret_cast<const float4 *>(plane + g.b10);
        r.t11 = *reinterpret_cast<const float4 *>(plane + g.b11);
        if (HAS_MASK) r.mq = *reinterpret_cast<const float4 *>(quads + g.b00);
    }
}

// Channel-planar stack [S,4,H,W] - the reference's own tensor layout, what render_tgt_rgb_depth /
// render_novel_view_dynamic callers hand over.  A tap pair (x0, x0+1) of one channel row is ONE 8-byte load, so a plane costs 8
// buffer_load_dwordx2 (4 channels x north / south row) on a single VGPR offset: the channel and the south row go into the instruction's
// scalar offset.  The descriptor ends exactly where the tensor ends, so the east / south neighbour of the very last texel reads as 0
// instead of faulting; inside the tensor an out-of-image neighbour is some finite texel of the next row / channel, and its weight is
// exactly 0 (x0+1 == W only when the clamped coordinate is exactly W-1) - the same argument as for the tail-padded interleaved stack.
typedef unsigned mpf_v2u __attribute__((ext_vector_type(2)));
struct MpfPlanarSrc {                 // where plane s of a channel-planar stack lives: rgb = 3 channel planes, sigma = 1
    const char *rgb, *sigma;          // [S,4,H,W]: sigma = rgb + 12 N, both strides 16 N;  rgb [S,3,H,W] + sigma [S,1,H,W]: strides 12 N / 4 N
    size_t rgb_stride, sigma_stride;  // bytes between planes
    unsigned rgb_last, sigma_last;    // bytes from the LAST plane's start to the end of the respective tensor: its descriptor limit (reads past
                                      // it return 0); the east / south spill of an earlier plane lands in the next plane, i.e. in valid memory
};
template <bool HAS_MASK>
MPF_DEV void mpf_fetch_planar(const MpfPlanarSrc &src, const int sp, const bool last, const char *__restrict__ quads, const unsigned quad_span,
                              const MpfGeom &g, MpfRaw4 &r, const unsigned chan_bytes, const unsigned row_bytes)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src.rgb + (size_t)sp * src.rgb_stride), 0,
                                                                   last ? src.rgb_last : 0xFFFFFFFCu, 0x00020000);
    __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src.sigma + (size_t)sp * src.sigma_stride), 0,
                                                                   last ? src.sigma_last : 0xFFFFFFFCu, 0x00020000);
    const mpf_v2u n0 = __builtin_amdgcn_raw_buffer_load_b64(rs, g.b00, 0, 0), s0 = __builtin_amdgcn_raw_buffer_load_b64(rs, g.b00, row_bytes, 0);
    const mpf_v2u n1 = __builtin_amdgcn_raw_buffer_load_b64(rs, g.b00, chan_bytes, 0), s1 = __builtin_amdgcn_raw_buffer_load_b64(rs, g.b00, chan_bytes + row_bytes, 0);
    const mpf_v2u n2 = __builtin_amdgcn_raw_buffer_load_b64(rs, g.b00, 2u * chan_bytes, 0), s2 = __builtin_amdgcn_raw_buffer_load_b64(rs, g.b00, 2u * chan_bytes + row_bytes, 0);
    const mpf_v2u n3 = __builtin_amdgcn_raw_buffer_load_b64(rg, g.b00, 0, 0), s3 = __builtin_amdgcn_raw_buffer_load_b64(rg, g.b00, row_bytes, 0);
    if (HAS_MASK) {
        __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(quads), 0, quad_span, 0x00020000);
        r.mq = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rq, g.b01, 0, 0));
    }
    auto f = [](unsigned u) { return __builtin_bit_cast(float, u); };
    r.t00 = make_float4(f(n0.x), f(n1.x), f(n2.x), f(n3.x));
    r.t01 = make_float4(f(n0.y), f(n1.y), f(n2.y), f(n3.y));
    r.t10 = make_float4(f(s0.x), f(s1.x), f(s2.x), f(s3.x));
    r.t11 = make_float4(f(s0.y), f(s1.y), f(s2.y), f(s3.y));
}

template <class Geom>
MPF_DEV float mpf_tap4w(const Geom &g, float a, float b, float c, float d)
{
    float o = a * g.nw;
    o = fmaf(b, g.ne, o);
    o = fmaf(c, g.sw, o);
    o = fmaf(d, g.se, o);
    return o;
}

template <int NL, bool HAS_MASK, bool WANT_DEPTH = true>
struct MpfAcc {
    double acc;
    MpfCsum<NL> cw, cd, co, c0, c1, c2;
    float nvalid;
    MPF_DEV void init()
    {
        acc = 1.0; nvalid = 0.0f;
        cw.init(); cd.init(); co.init(); c0.init(); c1.init(); c2.init();
    }
    // composite plane s (geometry g, fetched taps r) given the distance to plane s+1
    template <class Geom>
    MPF_DEV void step(const Geom &g, const MpfRaw4 &r, float dist, int s)
    {
        float cr = mpf_tap4w(g, r.t00.x, r.t01.x, r.t10.x, r.t11.x);
        float cg = mpf_tap4w(g, r.t00.y, r.t01.y, r.t10.y, r.t11.y);
        float cb = mpf_tap4w(g, r.t00.z, r.t01.z, r.t10.z, r.t11.z);
        float sg = mpf_tap4w(g, r.t00.w, r.t01.w, r.t10.w, r.t11.w);
        sg = (g.Z >= 0.0f) ? sg : 0.0f;
        float T = mpf_expf_fast(-sg * dist);
        float alpha = 1.0f - T;
        float tacc = (float)acc;
        float w = tacc * alpha;
        acc *= (double)(T + 1e-6f);
        c0.push(w * cr); c1.push(w * cg); c2.push(w * cb);
        if (WANT_DEPTH) { cw.push(w); cd.push(w * g.Z); }
        if (HAS_MASK) co.push(w * mpf_tap4w(g, r.mq.x, r.mq.y, r.mq.z, r.mq.w));
        if (((s + 1) & 15) == 0) {
            c0.fold(s + 1); c1.fold(s + 1); c2.fold(s + 1);
            if (WANT_DEPTH) { cw.fold(s + 1); cd.fold(s + 1); }
            if (HAS_MASK) co.fold(s + 1);
        }
    }
};

// DBG (bench-only ablations, results are NOT valid): 1 = no gathers (taps synthesised from the geometry), 2 = gathers and
// bilinear sums only (geometry of plane 0 reused for every plane, no distance / exp / composite)
template <bool HAS_MASK, int NL, int TW, int TH, bool KS, bool TP, int DBG = 0, bool AUX = true, bool PLANAR = false>
MPF_DEV void mpf_wc2_body(const float *__restrict__ rgba, const float *__restrict__ quads, const float *__restrict__ params_global,
                          int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                          float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out,
                          const unsigned tile, const MpfPlanarSrc *planar_src = nullptr)
{
    const MpfConstParams params = (MpfConstParams)params_global;
    const int64_t N = (int64_t)H * W;
    const unsigned tiles_x = (W + TW - 1) / TW;
    const int x = (tile % tiles_x) * TW + (threadIdx.x % TW);
    const int y = (tile / tiles_x) * TH + (threadIdx.x / TW);
    const bool active = (x < W) && (y < H);
    MpfConsts c;
    c.fx = (float)min(x, W - 1); c.fy = (float)min(y, H - 1);
    c.W = W; c.H = H; c.Wf = (float)W; c.Hf = (float)H;
    c.halfW = (float)W * 0.5f; c.halfH = (float)H * 0.5f;
    c.rhalfW = 1.0f / c.halfW; c.rhalfH = 1.0f / c.halfH;      // IEEE division (build flag): RN(1/d), which mpf_div_by_const relies on
    c.maxx = (float)(W - 1); c.maxy = (float)(H - 1);
    c.row_bytes = (unsigned)W * 16u;
    mpf_consts_pose(c, params);
    const char *qbase = reinterpret_cast<const char *>(quads);
    const char *pbase = reinterpret_cast<const char *>(rgba);
    const size_t plane_bytes = (size_t)N * 16;
    const unsigned span = (unsigned)(N + W + 1) * 16u;        // a plane plus the row / texel the south-east taps may touch (next plane or tail padding)

    MpfAcc<NL, HAS_MASK, AUX> A;
    A.init();
    MpfGeom ga, gb;
    MpfRaw4 ra, rb;
    A.nvalid += mpf_geom<KS, TP, AUX, PLANAR>(params, 0, c, ga);
    if (DBG == 1) {
        for (int s = 0; s < S; ++s) {
            A.nvalid += mpf_geom<KS, TP, AUX>(params, min(s + 1, S - 1), c, gb);
            ra.t00 = make_float4(ga.nw, ga.ne, ga.sw, ga.se); ra.t01 = make_float4(ga.X, ga.Y, ga.Z, ga.nw);
            ra.t10 = ra.t00; ra.t11 = ra.t01; ra.mq = ra.t00;
            A.step(ga, ra, mpf_norm3_nr(gb.X - ga.X, gb.Y - ga.Y, gb.Z - ga.Z), s);
            ga = gb;
        }
    } else if (DBG == 5 || DBG == 6) {   // t00, t10, quad for all lanes; t01, t11 only for every 10th lane + row ends (5) / never (6)
        float acc4 = 0.0f;
        const unsigned lane = threadIdx.x & 63u;
        const bool needy = (DBG == 5) && ((lane % 10u) == 0u || (lane & 31u) == 31u);
        for (int s = 0; s < S; ++s) {
            const char *pl = pbase + (size_t)s * plane_bytes;
            float4 a = *reinterpret_cast<const float4 *>(pl + ga.b00);
            float4 b = *reinterpret_cast<const float4 *>(pl + ga.b10);
            float4 q = *reinterpret_cast<const float4 *>(qbase + ga.b00);
            acc4 += a.x * ga.nw + a.y * ga.ne + a.z * ga.sw + a.w * ga.se + b.x * ga.nw + b.y * ga.ne + b.z * ga.sw + b.w * ga.se + q.x + q.y + q.z + q.w;
            if (needy) {
                float4 c2 = *reinterpret_cast<const float4 *>(pl + ga.b01);
                float4 d2 = *reinterpret_cast<const float4 *>(pl + ga.b11);
                acc4 += c2.x * ga.nw + c2.y * ga.ne + c2.z * ga.sw + c2.w * ga.se + d2.x + d2.y + d2.z + d2.w;
            }
        }
        A.c0.a[0] = acc4;
    } else if (DBG == 3 || DBG == 4) {   // 3: one 16-byte gather per plane (t00 only); 4: two (t00, t10)
        float acc4 = 0.0f;
        for (int s = 0; s < S; ++s) {
            const char *pl = pbase + (size_t)s * plane_bytes;
            float4 a = *reinterpret_cast<const float4 *>(pl + ga.b00);
            acc4 += a.x * ga.nw + a.y * ga.ne + a.z * ga.sw + a.w * ga.se;
            if (DBG == 4) {
                float4 b = *reinterpret_cast<const float4 *>(pl + ga.b10);
                acc4 += b.x * ga.nw + b.y * ga.ne + b.z * ga.sw + b.w * ga.se;
            }
        }
        A.c0.a[0] = acc4;
    } else if (DBG == 2) {
        float acc4 = 0.0f;
        for (int s = 0; s < S; ++s) {
            mpf_fetch2<HAS_MASK, TP>(pbase + (size_t)s * plane_bytes, qbase, ga, ra, c.row_bytes, span);
            acc4 += mpf_tap4w(ga, ra.t00.x, ra.t01.x, ra.t10.x, ra.t11.x) + mpf_tap4w(ga, ra.t00.y, ra.t01.y, ra.t10.y, ra.t11.y) +
                    mpf_tap4w(ga, ra.t00.z, ra.t01.z, ra.t10.z, ra.t11.z) + mpf_tap4w(ga, ra.t00.w, ra.t01.w, ra.t10.w, ra.t11.w);
            if (HAS_MASK) acc4 += mpf_tap4w(ga, ra.mq.x, ra.mq.y, ra.mq.z, ra.mq.w);
        }
        A.c0.a[0] = acc4;
    } else {
    // PLANAR: the stack is channel-planar as the reference holds it (planar_src says where: one [S,4,H,W] tensor, or rgb + sigma tensors)
    const unsigned chan_bytes = (unsigned)N * 4u, row4 = (unsigned)W * 4u, quad_span = (unsigned)N * 16u;
    auto FETCH = [&](int sp, const MpfGeom &g, MpfRaw4 &r) {
        if (PLANAR) {
            mpf_fetch_planar<HAS_MASK>(*planar_src, sp, sp + 1 == S, qbase, quad_span, g, r, chan_bytes, row4);
        } else {
            mpf_fetch2<HAS_MASK, TP>(pbase + (size_t)sp * plane_bytes, qbase, g, r, c.row_bytes, span);
        }
    };
    FETCH(0, ga, ra);

    int s = 0;
    while (s + 2 < S) {
        A.nvalid += mpf_geom<KS, TP, AUX, PLANAR>(params, s + 1, c, gb);
        FETCH(s + 1, gb, rb);
        A.step(ga, ra, mpf_norm3_nr(gb.X - ga.X, gb.Y - ga.Y, gb.Z - ga.Z), s);
        A.nvalid += mpf_geom<KS, TP, AUX, PLANAR>(params, s + 2, c, ga);
        FETCH(s + 2, ga, ra);
        A.step(gb, rb, mpf_norm3_nr(ga.X - gb.X, ga.Y - gb.Y, ga.Z - gb.Z), s + 1);
        s += 2;
    }
    if (s + 1 < S) {
        A.nvalid += mpf_geom<KS, TP, AUX, PLANAR>(params, s + 1, c, gb);
        FETCH(s + 1, gb, rb);
        A.step(ga, ra, mpf_norm3_nr(gb.X - ga.X, gb.Y - ga.Y, gb.Z - ga.Z), s);
        A.step(gb, rb, 1e3f, s + 1);
    } else {
        A.step(ga, ra, 1e3f, s);
    }
    }
    if (active) {
        const int64_t n = (int64_t)y * W + x;
        const float fr = A.c0.final(), fg = A.c1.final(), fb = A.c2.final();
        rgb_out[n] = fr;
        rgb_out[N + n] = fg;
        rgb_out[2 * N + n] = fb;
        if (u8_out) { u8_out[3 * n] = mpf_to_u8(fb); u8_out[3 * n + 1] = mpf_to_u8(fg); u8_out[3 * n + 2] = mpf_to_u8(fr); }   // BGR, utils/utils.py:240-242
        if (AUX && depth_out) depth_out[n] = A.cd.final() / (A.cw.final() + 1e-5f);
        if (HAS_MASK) om_out[n] = A.co.final();
        if (AUX && tgt_mask_out) tgt_mask_out[n] = A.nvalid;
    }
}

#ifdef MPF_WITNESS            // timing ablations of the Stage B body (results invalid): witness build only
template <bool HAS_MASK, int TW, int TH, int DBG>
__global__ void __launch_bounds__(TW *TH, 4)
k_warp_composite_dbg(const float *__restrict__ rgba, const float *__restrict__ quads, const float *__restrict__ params,
                     int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                     float *__restrict__ om_out, float *__restrict__ tgt_mask_out)
{
    mpf_wc2_body<HAS_MASK, 2, TW, TH, true, true, DBG>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, nullptr,
                                                       mpf_xcd_remap(blockIdx.x, gridDim.x));
}
#endif

// Tile order: strips of MPF_STRIP_TILES tiles wide, row by row inside a strip.  With the image-wide row-major order the ~80 tiles an
// XCD has resident at a time form a band 2-3 tile rows tall and as wide as the image; the two poses of a pair displace a tile's source
// footprint by up to ~100 px in BOTH directions, so the views of a multi-view launch rarely met in that XCD's L2.  A 4-tile-wide strip
// makes the resident set a compact 128 x 160 px patch: one launch for two views went 160 -> 146 us per view at 64x640x960 (four views
// 148 -> 132), widths 2 / 3 / 4 / 6 / 8: 155.5 / 149.8 / 146.1 / 152.5 / 147.3 (profiles/r2/stage_b_strip_order.log).  A pure
// scheduling choice: results never depend on it.
#ifndef MPF_STRIP_TILES
#define MPF_STRIP_TILES 4
#endif
// mpf_xcd_remap for a role that owns only NX of the 8 XCDs (k_pair_overlap with the roles partitioned by XCD): j = k * NX + x is the k-th
// block of the role's x-th XCD; that XCD gets the x-th contiguous chunk of the logical order.
MPF_DEV unsigned mpf_xcd_remap_n(unsigned j, unsigned nwg, unsigned NX)
{
    const unsigned x = j % NX, k = j / NX, q = nwg / NX, r = nwg % NX;
    return ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

MPF_DEV unsigned mpf_strip_order(unsigned t, unsigned tiles_x, unsigned tiles_y)
{
    const unsigned SWd = MPF_STRIP_TILES, nfull = tiles_x / SWd, full = nfull * SWd * tiles_y;
    if (t < full) {
        const unsigned strip = t / (SWd * tiles_y), r = t - strip * SWd * tiles_y;
        return (r / SWd) * tiles_x + strip * SWd + (r % SWd);
    }
    const unsigned rem = tiles_x - nfull * SWd, r = t - full;              // the last, narrower strip
    return (r / rem) * tiles_x + nfull * SWd + (r % rem);
}

template <bool HAS_MASK, int NL, int TW, int TH, bool TP>
MPF_DEV void mpf_wc2_select(const float *__restrict__ rgba, const float *__restrict__ quads, const float *__restrict__ params,
                            int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                            float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out,
                            const unsigned tile)
{
    const bool pinhole = (params[1] == 0.0f) & (params[3] == 0.0f) & (params[6] == 0.0f) & (params[7] == 0.0f) & (params[8] == 1.0f);
    const bool aux = (depth_out != nullptr) | (tgt_mask_out != nullptr);   // depth / validity count wanted at all?
    if (pinhole && !aux)
        mpf_wc2_body<HAS_MASK, NL, TW, TH, true, TP, 0, false>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile);
    else if (pinhole)
        mpf_wc2_body<HAS_MASK, NL, TW, TH, true, TP, 0, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile);
    else
        mpf_wc2_body<HAS_MASK, NL, TW, TH, false, TP, 0, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile);
}

// the same body on the reference's own channel-planar [S,4,H,W] stack (mpf_warp_composite(interleaved = 0): what a caller of
// render_tgt_rgb_depth / render_novel_view_dynamic, utils/mpi/mpi_rendering.py:259-349, hands over) - same tile shape, order and occupancy
template <bool HAS_MASK, int NL>
__global__ void __launch_bounds__(256, 5)
k_warp_composite_planar(const MpfPlanarSrc src, const float *__restrict__ quads, const float *__restrict__ params,
                        int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                        float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out)
{
    constexpr int TW = 32, TH = 8;
    const unsigned tile = mpf_strip_order(mpf_xcd_remap(blockIdx.x, gridDim.x), (W + TW - 1) / TW, (H + TH - 1) / TH);
    const MpfConstParams cp = (MpfConstParams)params;
    const bool pinhole = (cp[1] == 0.0f) & (cp[3] == 0.0f) & (cp[6] == 0.0f) & (cp[7] == 0.0f) & (cp[8] == 1.0f);
    const bool aux = (depth_out != nullptr) | (tgt_mask_out != nullptr);
    const float *rgba = reinterpret_cast<const float *>(src.rgb);
    if (pinhole && !aux)
        mpf_wc2_body<HAS_MASK, NL, TW, TH, true, false, 0, false, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, &src);
    else if (pinhole)
        mpf_wc2_body<HAS_MASK, NL, TW, TH, true, false, 0, true, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, &src);
    else
        mpf_wc2_body<HAS_MASK, NL, TW, TH, false, false, 0, true, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, &src);
}

template <bool HAS_MASK, int NL, int TW, int TH, int WPS, bool TP>
__global__ void __launch_bounds__(TW *TH, WPS)
k_warp_composite_v2(const float *__restrict__ rgba, const float *__restrict__ quads, const float *__restrict__ params,
                    int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                    float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out)
{
    mpf_wc2_select<HAS_MASK, NL, TW, TH, TP>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out,
                                             mpf_strip_order(mpf_xcd_remap(blockIdx.x, gridDim.x), (W + TW - 1) / TW, (H + TH - 1) / TH));
}

// Several views of ONE stack in one launch (the reference renders two poses of every stack, utils/utils.py:210-236, and
// `repeat` such pairs per image, gen_3dphoto_dynamic_v2.py:99-118).  Logical block l = tile * V + view: the V workgroups of a
// tile are dispatched back to back on the same XCD, walk the planes at the same pace and so find each other's texels in
// that XCD's L2 (or in the Infinity Cache) - the 16*S*N-byte stack crosses the HBM interface once per launch instead of
// once per view.  Same body, same registers, same occupancy as the single-view kernel; results are bit-identical.
struct MpfViewSet { MpfWarpView v[MPF_MAX_VIEWS]; };

template <bool HAS_MASK, int NL, int TW, int TH, int WPS, bool TP>
__global__ void __launch_bounds__(TW *TH, WPS)
k_warp_composite_views(const float *__restrict__ rgba, const MpfViewSet vs, const unsigned V, int S, int H, int W, const unsigned view_shift)
{
    const unsigned l = mpf_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned view = l % V;
    // view_shift (mpf_tune("view_shift", k), default 8): the odd views walk the strip-major tile sequence k positions ahead of the even ones
    // (see g_view_shift).  A rotation of the sequence is a bijection: every tile of every view is still rendered exactly once.
    const unsigned tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, ntiles = tiles_x * tiles_y;
    const unsigned seq = (view & 1u) ? (l / V + view_shift) % ntiles : l / V;
    const unsigned tile = mpf_strip_order(seq, tiles_x, tiles_y);
    const MpfWarpView &w = vs.v[view];
    mpf_wc2_select<HAS_MASK, NL, TW, TH, TP>(rgba, w.d_mask_quads, w.d_params, S, H, W, w.d_rgb, w.d_depth, w.d_objmask, w.d_tgt_mask,
                                             w.d_rgb_u8_bgr, tile);
}

// ---------------------------------------------------------------------------------------------------------------
// Stage B, LDS-staged variant ("source-pixel neighbourhoods through LDS", tail-padded interleaved stack, S <= 256)
//
// The gather kernels issue 4 x 16-byte gathers per pixel and plane; neighbouring pixels re-fetch each other's texels
// through the L1 / texture-address path (TA busy ~55 %, 3.07e6 VMEM instructions per launch at 64x640x960).  Here a 32x8
// target tile's source FOOTPRINT on plane s (a bounding box of at most 48 x 16 texels) is fetched once per workgroup with
// fully coalesced 16-byte buffer loads (<= 3 per wave and plane, usually 2.25), parked in LDS (two 12 KB buffers,
// ping-pong, one barrier per plane), and every pixel takes its four taps from there with ds_read_b128 at immediate
// offsets.  Same arithmetic (mpf_geom_core, MpfAcc), bit-identical results.
//   * the footprint boxes of all S planes are computed once per workgroup from the tile's 4 corner pixels (a homography
//     maps the tile's extremes to its corners; +-1 texel of margin covers rounding), S*4 corner evaluations on 256 threads
//   * a plane whose box does not fit (extreme poses) or whose denominator comes near 0 on the tile falls back to direct
//     gathers for that plane - a workgroup-uniform branch, so correctness never depends on the pose
//   * staging costs no VALU: a thread's global offset (row*W + col)*16 and LDS slot are loop invariants, the box origin
//     goes into the buffer instruction's scalar offset, out-of-box lanes are fetched too (in range: the buffer descriptor
//     clips at the end of the padded plane) and simply never read
// ---------------------------------------------------------------------------------------------------------------
#define MPF_LT_PITCH 48
#define MPF_LT_ROWS 16
#define MPF_LT_TEXELS (MPF_LT_PITCH * MPF_LT_ROWS)     // 768 texels = 12 KB per buffer = 3 passes of 256 threads
#define MPF_LT_PASSES 3
#define MPF_LT_MAXS 256

struct MpfGeomL {
    float nw, ne, sw, se;
    float X, Y, Z;
    unsigned t;          // y0 * MPF_LT_PITCH + x0 : position in LDS pitch units (box origin subtracted at use)
    unsigned b00;        // (y0 * W + x0) * 16     : byte offset inside a plane (mask quads, direct fallback)
};

struct MpfBox {          // wave-uniform (SGPRs)
    unsigned goff;       // (ymin * W + xmin) * 16 : byte offset of the box origin inside a plane
    unsigned lorg;       // (ymin * MPF_LT_PITCH + xmin) * 16
    unsigned h;          // rows staged; 0 = this plane is not staged (direct gathers)
};



template <bool KS, bool AUX>
MPF_DEV float mpf_geom_l(MpfConstParams params, int s, const MpfConsts &c, MpfGeomL &g)
{
    int x0, y0;
    float qz;
    const float valid = mpf_geom_core<KS, AUX>(params, s, c, g.nw, g.ne, g.sw, g.se, g.X, g.Y, g.Z, x0, y0, qz);
    g.t = __umul24((unsigned)y0, (unsigned)MPF_LT_PITCH) + (unsigned)x0;
    g.b00 = (__umul24((unsigned)y0, (unsigned)c.W) + (unsigned)x0) * 16u;
    return valid;
}

// PLANAR: the stack is channel-planar as the reference holds it ([S,4,H,W], or rgb [S,3,H,W] + sigma [S,1,H,W]: planar_src).  The footprint
// box is fetched with coalesced DWORD loads from the four channel planes (same box raster, one texel per lane and pass) and interleaved
// on the way into LDS, so the tap reads are the same four ds_read_b128 - what costs the planar gather kernel 8 + 1 uncoalesced gathers per
// plane and pixel (TA-bound, 0.32-0.35 of the HBM roofline) becomes <= 12 coalesced 4-byte loads per lane and plane: 248 -> 236 us with a mask,
// 226 -> 214 without at 64 x 640 x 960 (profiles/r4/stage_b_planar_lds.log).  Tried on top and dropped: the mask's footprint through LDS as
// well (97 VGPRs + 52 spilled SGPRs, four unaligned ds_read_b32 per pixel: 289 us), 5 workgroups per CU (223 us without a mask), and the box rows as
// 16-byte loads of one channel into a channel-planar LDS tile (3 instead of 12 loads per lane and plane, taps = 8 ds_read2_b32: 244 / 215 us) -
// the LDS forms are bound by the per-plane barrier and the LDS round trip, not by the global loads.
template <bool HAS_MASK, int NL, bool KS, bool AUX, bool PLANAR = false>
MPF_DEV void mpf_wcl_body(const float *__restrict__ rgba, const float *__restrict__ quads, const float *__restrict__ params,
                          int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                          float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out,
                          const unsigned tile, float4 *s_tex, uint2 *s_box, const MpfPlanarSrc *planar_src = nullptr)
{
    constexpr int TW = 32, TH = 8;
    const int64_t N = (int64_t)H * W;
    const unsigned tiles_x = (W + TW - 1) / TW;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const unsigned tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = tx0 + (int)(tid % TW);
    const int y = ty0 + (int)(tid / TW);
    const bool active = (x < W) && (y < H);
    MpfConsts c;
    c.fx = (float)min(x, W - 1); c.fy = (float)min(y, H - 1);
    c.W = W; c.H = H; c.Wf = (float)W; c.Hf = (float)H;
    c.halfW = (float)W * 0.5f; c.halfH = (float)H * 0.5f;
    c.rhalfW = 1.0f / c.halfW; c.rhalfH = 1.0f / c.halfH;      // IEEE division (build flag): RN(1/d), which mpf_div_by_const relies on
    c.maxx = (float)(W - 1); c.maxy = (float)(H - 1);
    c.row_bytes = (unsigned)W * 16u;
    mpf_consts_pose(c, (MpfConstParams)params);
    const char *qbase = reinterpret_cast<const char *>(quads);
    const char *pbase = reinterpret_cast<const char *>(rgba);
    const size_t plane_bytes = (size_t)N * 16;
    char *lds = reinterpret_cast<char *>(s_tex);

    // ---- footprint boxes of this tile on every plane --------------------------------------------------------------
    int unfit = 0;
    {
        const int xa = min(tx0, W - 1), xb = min(tx0 + TW - 1, W - 1);
        const int ya = min(ty0, H - 1), yb = min(ty0 + TH - 1, H - 1);
        for (unsigned e = tid; e < (unsigned)S * 4u; e += TW * TH) {
            const unsigned sp = e >> 2, k = e & 3u;
            MpfConsts cc = c;
            cc.fx = (float)((k & 1u) ? xb : xa);
            cc.fy = (float)((k & 2u) ? yb : ya);
            float w0, w1, w2, w3, X, Y, Z, qz;
            int cx, cy;
            mpf_geom_core<KS, false>(params, (int)sp, cc, w0, w1, w2, w3, X, Y, Z, cx, cy, qz);
            int xmn = cx, xmx = cx, ymn = cy, ymx = cy;
            float zmn = qz;
#pragma unroll
            for (int m = 1; m <= 2; m <<= 1) {          // the 4 corners sit in 4 adjacent lanes
                xmn = min(xmn, __shfl_xor(xmn, m)); xmx = max(xmx, __shfl_xor(xmx, m));
                ymn = min(ymn, __shfl_xor(ymn, m)); ymx = max(ymx, __shfl_xor(ymx, m));
                zmn = fminf(zmn, __shfl_xor(zmn, m));
            }
            xmn = max(xmn - 1, 0); xmx = min(xmx + 2, W);      // +1 east tap, +-1 margin; column W / row H = the padding
            ymn = max(ymn - 1, 0); ymx = min(ymx + 2, H);
            const int bw = xmx - xmn + 1, bh = ymx - ymn + 1;
            const bool ok = (bw <= MPF_LT_PITCH) && (bh <= MPF_LT_ROWS) && (zmn > 0.0625f);
            unfit |= ok ? 0 : 1;
            if (k == 0)
                s_box[sp] = make_uint2((unsigned)ymn * (unsigned)W + (unsigned)xmn, ((unsigned)ymn * MPF_LT_PITCH + (unsigned)xmn) | ((unsigned)bh << 24));
        }
    }
    // a plane whose footprint does not fit the LDS tile (extreme pose), or a denominator near 0 on the tile: the whole
    // workgroup takes the gather path for this tile (uniform; the barrier doubles as the one publishing s_box)
    if (__syncthreads_or(unfit | ((H >= 65536) | (W >= 65536)))) {
        if (PLANAR) mpf_wc2_body<HAS_MASK, NL, TW, TH, KS, false, 0, AUX, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, planar_src);
        else mpf_wc2_body<HAS_MASK, NL, TW, TH, KS, true, 0, AUX>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile);
        return;
    }

    auto box = [&](int s) -> MpfBox {
        const uint2 b = s_box[s];
        const unsigned b0 = __builtin_amdgcn_readfirstlane(b.x), b1 = __builtin_amdgcn_readfirstlane(b.y);
        MpfBox r;
        r.goff = b0 * 16u;
        r.lorg = (b1 & 0xFFFFFFu) * 16u;
        r.h = b1 >> 24;
        return r;
    };
    // a thread's staging slots: texel idx = tid + 256 k of the 48-pitch box raster
    unsigned gconst[MPF_LT_PASSES];
#pragma unroll
    for (int k = 0; k < MPF_LT_PASSES; ++k) {
        const unsigned idx = tid + (unsigned)(TW * TH) * k;
        gconst[k] = ((idx / MPF_LT_PITCH) * (unsigned)W + (idx % MPF_LT_PITCH)) * 16u;
    }
    // one descriptor per plane: [plane base, +(N + W + 1) texels) - row H / column W of the last plane are the tail padding
    const unsigned span = (unsigned)(N + W + 1) * 16u;
    mpf_v4u L[MPF_LT_PASSES];
    const unsigned chan_bytes = (unsigned)N * 4u;
    auto issue = [&](int s, const MpfBox &b) {
        if (PLANAR) {
            // one descriptor per plane and tensor; it ends where the TENSOR ends for the last plane (reads past it return 0, their weight is 0),
            // earlier planes spill into the next channel / plane, i.e. into valid memory (same argument as mpf_fetch_planar)
            const bool last = s + 1 == S;
            __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(planar_src->rgb + (size_t)s * planar_src->rgb_stride), 0,
                                                                           last ? planar_src->rgb_last : 0xFFFFFFFCu, 0x00020000);
            __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(planar_src->sigma + (size_t)s * planar_src->sigma_stride), 0,
                                                                           last ? planar_src->sigma_last : 0xFFFFFFFCu, 0x00020000);
            const unsigned g4 = b.goff >> 2;                  // (ymin * W + xmin) * 4: byte offset of the box origin inside one channel plane
#pragma unroll
            for (int k = 0; k < MPF_LT_PASSES; ++k)
                if ((unsigned)(TW * TH) * k + 64u * wave < MPF_LT_PITCH * b.h) {
                    const unsigned v = gconst[k] >> 2;
                    L[k].x = __builtin_amdgcn_raw_buffer_load_b32(rc, v, g4, 0);
                    L[k].y = __builtin_amdgcn_raw_buffer_load_b32(rc, v, g4 + chan_bytes, 0);
                    L[k].z = __builtin_amdgcn_raw_buffer_load_b32(rc, v, g4 + 2u * chan_bytes, 0);
                    L[k].w = __builtin_amdgcn_raw_buffer_load_b32(rg, v, g4, 0);
                }
            return;
        }
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(pbase + (size_t)s * plane_bytes), 0, span, 0x00020000);
#pragma unroll
        for (int k = 0; k < MPF_LT_PASSES; ++k)
            if ((unsigned)(TW * TH) * k + 64u * wave < MPF_LT_PITCH * b.h)
                L[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, gconst[k], b.goff, 0);
    };
    auto stage = [&](const MpfBox &b, const unsigned buf) {
#pragma unroll
        for (int k = 0; k < MPF_LT_PASSES; ++k)
            if ((unsigned)(TW * TH) * k + 64u * wave < MPF_LT_PITCH * b.h)
                *reinterpret_cast<mpf_v4u *>(lds + buf * (MPF_LT_TEXELS * 16) + (tid + (unsigned)(TW * TH) * k) * 16u) = L[k];
    };
    auto taps = [&](const MpfBox &b, const MpfGeomL &g, const unsigned buf, MpfRaw4 &r) {
        const char *a = lds + ((g.t << 4) + (buf * (MPF_LT_TEXELS * 16) - b.lorg));
        r.t00 = *reinterpret_cast<const float4 *>(a);
        r.t01 = *reinterpret_cast<const float4 *>(a + 16);
        r.t10 = *reinterpret_cast<const float4 *>(a + MPF_LT_PITCH * 16);
        r.t11 = *reinterpret_cast<const float4 *>(a + MPF_LT_PITCH * 16 + 16);
    };

    MpfConstParams cparams = (MpfConstParams)params;
    MpfAcc<NL, HAS_MASK, AUX> A;
    A.init();
    MpfGeomL ga, gb;
    MpfRaw4 ra, rb;                                      // t00..t11 are filled from LDS in the step that uses them; .mq (a global gather) one step ahead
    MpfBox bA = box(0), bB = box(min(1, S - 1));
    issue(0, bA);
    stage(bA, 0u);
    if (S > 1) issue(1, bB);
    A.nvalid += mpf_geom_l<KS, AUX>(cparams, 0, c, ga);
    if (HAS_MASK) ra.mq = *reinterpret_cast<const float4 *>(qbase + ga.b00);
    __syncthreads();

    // plane s: geometry G_, box B_, LDS buffer BUF_, taps R_; its successor: geometry GN_, box BN_, taps RN_, loads in flight in L;
    // once they are parked in the other buffer, L takes the loads of plane s+2 (box BF_, read here)
#define MPF_WCL_STEP(s_, G_, GN_, B_, BN_, BF_, BUF_, R_, RN_)                                                        \
    {                                                                                                                 \
        taps(B_, G_, BUF_, R_);                                                                                       \
        A.nvalid += mpf_geom_l<KS, AUX>(cparams, (s_) + 1, c, GN_);                                                   \
        if (HAS_MASK) RN_.mq = *reinterpret_cast<const float4 *>(qbase + GN_.b00);                                    \
        const float dist_ = mpf_norm3_nr(GN_.X - G_.X, GN_.Y - G_.Y, GN_.Z - G_.Z);                                   \
        A.step(G_, R_, dist_, (s_));                                                                                  \
        stage(BN_, 1u - (BUF_));                                                                                      \
        if ((s_) + 2 < S) { BF_ = box((s_) + 2); issue((s_) + 2, BF_); }                                              \
        __syncthreads();                                                                                              \
    }
    int s = 0;
    while (s + 2 < S) {
        MPF_WCL_STEP(s, ga, gb, bA, bB, bA, 0u, ra, rb)            // plane s   (buffer 0): parks s+1 in buffer 1, fetches s+2
        MPF_WCL_STEP(s + 1, gb, ga, bB, bA, bB, 1u, rb, ra)        // plane s+1 (buffer 1): parks s+2 in buffer 0, fetches s+3
        s += 2;
    }
    if (s + 1 < S) {
        MPF_WCL_STEP(s, ga, gb, bA, bB, bA, 0u, ra, rb)
        taps(bB, gb, 1u, rb);
        A.step(gb, rb, 1e3f, s + 1);
    } else {
        taps(bA, ga, 0u, ra);
        A.step(ga, ra, 1e3f, s);
    }
#undef MPF_WCL_STEP
    if (active) {
        const int64_t n = (int64_t)y * W + x;
        const float fr = A.c0.final(), fg = A.c1.final(), fb = A.c2.final();
        rgb_out[n] = fr;
        rgb_out[N + n] = fg;
        rgb_out[2 * N + n] = fb;
        if (u8_out) { u8_out[3 * n] = mpf_to_u8(fb); u8_out[3 * n + 1] = mpf_to_u8(fg); u8_out[3 * n + 2] = mpf_to_u8(fr); }
        if (AUX && depth_out) depth_out[n] = A.cd.final() / (A.cw.final() + 1e-5f);
        if (HAS_MASK) om_out[n] = A.co.final();
        if (AUX && tgt_mask_out) tgt_mask_out[n] = A.nvalid;
    }
}

#ifdef MPF_WITNESS            // rejected variants kept as witnesses (bit-identical, slower): the wave-private planar form and the LDS-staged interleaved form
// WAVE-PRIVATE footprints (round 5; mpf_tune("planar_lds", 2)): the 32 x 8 tile as four 32 x 2 wave strips.  Every wave computes the boxes of ITS strip
// (<= 48 x 6 texels), stages them into its own slab of LDS and reads its taps from there: the LDS traffic of a wave is ordered by the hardware, so
// the plane loop needs NO workgroup barrier - what the workgroup-wide box pays once per plane.  The price: the four strips' row halos overlap
// (4 x (2 + 3) source rows instead of 8 + 3), i.e. up to 20 instead of 12 coalesced dword loads per lane and plane.  Same arithmetic, bit-identical.
#define MPF_WT_ROWS 6
#define MPF_WT_TEXELS (MPF_LT_PITCH * MPF_WT_ROWS)     // 288 texels = 4.5 passes of 64 lanes
#define MPF_WT_PASSES 5
#define MPF_WT_MAXS 96

template <bool HAS_MASK, int NL, bool KS, bool AUX>
MPF_DEV void mpf_wcw_body(const float *__restrict__ quads, const float *__restrict__ params, int S, int H, int W, float *__restrict__ rgb_out,
                          float *__restrict__ depth_out, float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out,
                          const unsigned tile, float4 *s_tex, uint2 *s_box, const MpfPlanarSrc *planar_src)
{
    constexpr int TW = 32, TH = 8;
    const int64_t N = (int64_t)H * W;
    const unsigned tiles_x = (W + TW - 1) / TW;
    const int tx0 = (tile % tiles_x) * TW, ty0 = (tile / tiles_x) * TH;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x = tx0 + (int)(tid % TW);
    const int y = ty0 + (int)(tid / TW);
    const bool active = (x < W) && (y < H);
    MpfConsts c;
    c.fx = (float)min(x, W - 1); c.fy = (float)min(y, H - 1);
    c.W = W; c.H = H; c.Wf = (float)W; c.Hf = (float)H;
    c.halfW = (float)W * 0.5f; c.halfH = (float)H * 0.5f;
    c.rhalfW = 1.0f / c.halfW; c.rhalfH = 1.0f / c.halfH;
    c.maxx = (float)(W - 1); c.maxy = (float)(H - 1);
    c.row_bytes = (unsigned)W * 16u;
    mpf_consts_pose(c, (MpfConstParams)params);
    const char *qbase = reinterpret_cast<const char *>(quads);
    char *lds = reinterpret_cast<char *>(s_tex) + wave * (2 * MPF_WT_TEXELS * 16);          // this wave's two buffers
    uint2 *wbox = s_box + wave * MPF_WT_MAXS;

    // ---- footprint boxes of this WAVE's strip (rows 2 wave, 2 wave + 1 of the tile) on every plane
    int unfit = 0;
    {
        const int xa = min(tx0, W - 1), xb = min(tx0 + TW - 1, W - 1);
        const int ya = min(ty0 + 2 * (int)wave, H - 1), yb = min(ty0 + 2 * (int)wave + 1, H - 1);
        for (unsigned e = lane; e < (unsigned)S * 4u; e += 64u) {
            const unsigned sp = e >> 2, k = e & 3u;
            MpfConsts cc = c;
            cc.fx = (float)((k & 1u) ? xb : xa);
            cc.fy = (float)((k & 2u) ? yb : ya);
            float w0, w1, w2, w3, X, Y, Z, qz;
            int cx, cy;
            mpf_geom_core<KS, false>(params, (int)sp, cc, w0, w1, w2, w3, X, Y, Z, cx, cy, qz);
            int xmn = cx, xmx = cx, ymn = cy, ymx = cy;
            float zmn = qz;
#pragma unroll
            for (int m = 1; m <= 2; m <<= 1) {          // the 4 corners sit in 4 adjacent lanes
                xmn = min(xmn, __shfl_xor(xmn, m)); xmx = max(xmx, __shfl_xor(xmx, m));
                ymn = min(ymn, __shfl_xor(ymn, m)); ymx = max(ymx, __shfl_xor(ymx, m));
                zmn = fminf(zmn, __shfl_xor(zmn, m));
            }
            xmn = max(xmn - 1, 0); xmx = min(xmx + 2, W);
            ymn = max(ymn - 1, 0); ymx = min(ymx + 2, H);
            const int bw = xmx - xmn + 1, bh = ymx - ymn + 1;
            const bool ok = (bw <= MPF_LT_PITCH) && (bh <= MPF_WT_ROWS) && (zmn > 0.0625f);
            unfit |= ok ? 0 : 1;
            if (k == 0)
                wbox[sp] = make_uint2((unsigned)ymn * (unsigned)W + (unsigned)xmn, ((unsigned)ymn * MPF_LT_PITCH + (unsigned)xmn) | ((unsigned)bh << 24));
        }
    }
    // a strip whose footprint does not fit on some plane: the whole workgroup takes the gather path for this tile (uniform; the only barrier)
    if (__syncthreads_or(unfit | ((H >= 65536) | (W >= 65536)))) {
        mpf_wc2_body<HAS_MASK, NL, TW, TH, KS, false, 0, AUX, true>(reinterpret_cast<const float *>(planar_src->rgb), quads, params, S, H, W, rgb_out, depth_out, om_out,
                                                                    tgt_mask_out, u8_out, tile, planar_src);
        return;
    }
    auto box = [&](int s) -> MpfBox {
        const uint2 b = wbox[s];
        const unsigned b0 = __builtin_amdgcn_readfirstlane(b.x), b1 = __builtin_amdgcn_readfirstlane(b.y);
        MpfBox r;
        r.goff = b0 * 16u;
        r.lorg = (b1 & 0xFFFFFFu) * 16u;
        r.h = b1 >> 24;
        return r;
    };
    unsigned gconst[MPF_WT_PASSES];                          // a lane's staging slots: texel idx = lane + 64 k of the 48-pitch box raster
#pragma unroll
    for (int k = 0; k < MPF_WT_PASSES; ++k) {
        const unsigned idx = lane + 64u * k;
        gconst[k] = ((idx / MPF_LT_PITCH) * (unsigned)W + (idx % MPF_LT_PITCH)) * 4u;      // byte offset inside one channel plane
    }
    mpf_v4u L[MPF_WT_PASSES];
    const unsigned chan_bytes = (unsigned)N * 4u;
    auto issue = [&](int s, const MpfBox &b) {
        const bool last = s + 1 == S;
        __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(planar_src->rgb + (size_t)s * planar_src->rgb_stride), 0,
                                                                       last ? planar_src->rgb_last : 0xFFFFFFFCu, 0x00020000);
        __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(planar_src->sigma + (size_t)s * planar_src->sigma_stride), 0,
                                                                       last ? planar_src->sigma_last : 0xFFFFFFFCu, 0x00020000);
        const unsigned g4 = b.goff >> 2;
#pragma unroll
        for (int k = 0; k < MPF_WT_PASSES; ++k)
            if (64u * k < MPF_LT_PITCH * b.h) {                // wave-uniform: passes past the box's last row are skipped
                L[k].x = __builtin_amdgcn_raw_buffer_load_b32(rc, gconst[k], g4, 0);
                L[k].y = __builtin_amdgcn_raw_buffer_load_b32(rc, gconst[k], g4 + chan_bytes, 0);
                L[k].z = __builtin_amdgcn_raw_buffer_load_b32(rc, gconst[k], g4 + 2u * chan_bytes, 0);
                L[k].w = __builtin_amdgcn_raw_buffer_load_b32(rg, gconst[k], g4, 0);
            }
    };
    auto stage = [&](const MpfBox &b, const unsigned buf) {
#pragma unroll
        for (int k = 0; k < MPF_WT_PASSES; ++k)
            if (64u * k < MPF_LT_PITCH * b.h && (k < MPF_WT_PASSES - 1 || lane + 64u * k < MPF_WT_TEXELS))
                *reinterpret_cast<mpf_v4u *>(lds + buf * (MPF_WT_TEXELS * 16) + (lane + 64u * k) * 16u) = L[k];
    };
    auto taps = [&](const MpfBox &b, const MpfGeomL &g, const unsigned buf, MpfRaw4 &r) {
        const char *a = lds + ((g.t << 4) + (buf * (MPF_WT_TEXELS * 16) - b.lorg));
        r.t00 = *reinterpret_cast<const float4 *>(a);
        r.t01 = *reinterpret_cast<const float4 *>(a + 16);
        r.t10 = *reinterpret_cast<const float4 *>(a + MPF_LT_PITCH * 16);
        r.t11 = *reinterpret_cast<const float4 *>(a + MPF_LT_PITCH * 16 + 16);
    };

    MpfConstParams cparams = (MpfConstParams)params;
    MpfAcc<NL, HAS_MASK, AUX> A;
    A.init();
    MpfGeomL ga, gb;
    MpfRaw4 ra, rb;
    MpfBox bA = box(0), bB = box(min(1, S - 1));
    issue(0, bA);
    stage(bA, 0u);
    if (S > 1) issue(1, bB);
    A.nvalid += mpf_geom_l<KS, AUX>(cparams, 0, c, ga);
    if (HAS_MASK) ra.mq = *reinterpret_cast<const float4 *>(qbase + ga.b00);
    // no barrier anywhere below: a wave reads only what it wrote itself, and its LDS operations complete in order
#define MPF_WCW_STEP(s_, G_, GN_, B_, BN_, BF_, BUF_, R_, RN_)                                                        \
    {                                                                                                                 \
        taps(B_, G_, BUF_, R_);                                                                                       \
        A.nvalid += mpf_geom_l<KS, AUX>(cparams, (s_) + 1, c, GN_);                                                   \
        if (HAS_MASK) RN_.mq = *reinterpret_cast<const float4 *>(qbase + GN_.b00);                                    \
        const float dist_ = mpf_norm3_nr(GN_.X - G_.X, GN_.Y - G_.Y, GN_.Z - G_.Z);                                   \
        A.step(G_, R_, dist_, (s_));                                                                                  \
        stage(BN_, 1u - (BUF_));                                                                                      \
        if ((s_) + 2 < S) { BF_ = box((s_) + 2); issue((s_) + 2, BF_); }                                              \
    }
    int s = 0;
    while (s + 2 < S) {
        MPF_WCW_STEP(s, ga, gb, bA, bB, bA, 0u, ra, rb)
        MPF_WCW_STEP(s + 1, gb, ga, bB, bA, bB, 1u, rb, ra)
        s += 2;
    }
    if (s + 1 < S) {
        MPF_WCW_STEP(s, ga, gb, bA, bB, bA, 0u, ra, rb)
        taps(bB, gb, 1u, rb);
        A.step(gb, rb, 1e3f, s + 1);
    } else {
        taps(bA, ga, 0u, ra);
        A.step(ga, ra, 1e3f, s);
    }
#undef MPF_WCW_STEP
    if (active) {
        const int64_t n = (int64_t)y * W + x;
        const float fr = A.c0.final(), fg = A.c1.final(), fb = A.c2.final();
        rgb_out[n] = fr;
        rgb_out[N + n] = fg;
        rgb_out[2 * N + n] = fb;
        if (u8_out) { u8_out[3 * n] = mpf_to_u8(fb); u8_out[3 * n + 1] = mpf_to_u8(fg); u8_out[3 * n + 2] = mpf_to_u8(fr); }
        if (AUX && depth_out) depth_out[n] = A.cd.final() / (A.cw.final() + 1e-5f);
        if (HAS_MASK) om_out[n] = A.co.final();
        if (AUX && tgt_mask_out) tgt_mask_out[n] = A.nvalid;
    }
}

template <bool HAS_MASK, int NL>
__global__ void __launch_bounds__(256, 4)
k_warp_composite_planar_wave(const MpfPlanarSrc src, const float *__restrict__ quads, const float *__restrict__ params,
                             int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                             float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out)
{
    constexpr int TW = 32, TH = 8;
    const unsigned tile = mpf_strip_order(mpf_xcd_remap(blockIdx.x, gridDim.x), (W + TW - 1) / TW, (H + TH - 1) / TH);
    const MpfConstParams cp = (MpfConstParams)params;
    const bool pinhole = (cp[1] == 0.0f) & (cp[3] == 0.0f) & (cp[6] == 0.0f) & (cp[7] == 0.0f) & (cp[8] == 1.0f);
    const bool aux = (depth_out != nullptr) | (tgt_mask_out != nullptr);
    __shared__ float4 s_tex[4 * 2 * MPF_WT_TEXELS];
    __shared__ uint2 s_box[4 * MPF_WT_MAXS];
    if (pinhole && !aux)
        mpf_wcw_body<HAS_MASK, NL, true, false>(quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, s_tex, s_box, &src);
    else if (pinhole)
        mpf_wcw_body<HAS_MASK, NL, true, true>(quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, s_tex, s_box, &src);
    else
        mpf_wcw_body<HAS_MASK, NL, false, true>(quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, s_tex, s_box, &src);
}

template <bool HAS_MASK, int NL>
__global__ void __launch_bounds__(256, 4)
k_warp_composite_lds(const float *__restrict__ rgba, const MpfViewSet vs, const unsigned V, int S, int H, int W)
{
    const unsigned l = mpf_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned view = l % V, tile = l / V;
    const MpfWarpView &w = vs.v[view];
    const float *params = w.d_params;
    const bool pinhole = (params[1] == 0.0f) & (params[3] == 0.0f) & (params[6] == 0.0f) & (params[7] == 0.0f) & (params[8] == 1.0f);
    const bool aux = (w.d_depth != nullptr) | (w.d_tgt_mask != nullptr);
    __shared__ float4 s_tex[2 * MPF_LT_TEXELS];      // one set for the three bodies below
    __shared__ uint2 s_box[MPF_LT_MAXS];
    if (pinhole && !aux)
        mpf_wcl_body<HAS_MASK, NL, true, false>(rgba, w.d_mask_quads, params, S, H, W, w.d_rgb, w.d_depth, w.d_objmask, w.d_tgt_mask, w.d_rgb_u8_bgr, tile, s_tex, s_box);
    else if (pinhole)
        mpf_wcl_body<HAS_MASK, NL, true, true>(rgba, w.d_mask_quads, params, S, H, W, w.d_rgb, w.d_depth, w.d_objmask, w.d_tgt_mask, w.d_rgb_u8_bgr, tile, s_tex, s_box);
    else
        mpf_wcl_body<HAS_MASK, NL, false, true>(rgba, w.d_mask_quads, params, S, H, W, w.d_rgb, w.d_depth, w.d_objmask, w.d_tgt_mask, w.d_rgb_u8_bgr, tile, s_tex, s_box);
}
#endif   // MPF_WITNESS

template <bool HAS_MASK, int NL>
__global__ void __launch_bounds__(256, 4)
k_warp_composite_planar_lds(const MpfPlanarSrc src, const float *__restrict__ quads, const float *__restrict__ params,
                            int S, int H, int W, float *__restrict__ rgb_out, float *__restrict__ depth_out,
                            float *__restrict__ om_out, float *__restrict__ tgt_mask_out, uint8_t *__restrict__ u8_out)
{
    constexpr int TW = 32, TH = 8;
    const unsigned tile = mpf_strip_order(mpf_xcd_remap(blockIdx.x, gridDim.x), (W + TW - 1) / TW, (H + TH - 1) / TH);
    const MpfConstParams cp = (MpfConstParams)params;
    const bool pinhole = (cp[1] == 0.0f) & (cp[3] == 0.0f) & (cp[6] == 0.0f) & (cp[7] == 0.0f) & (cp[8] == 1.0f);
    const bool aux = (depth_out != nullptr) | (tgt_mask_out != nullptr);
    const float *rgba = reinterpret_cast<const float *>(src.rgb);
    __shared__ float4 s_tex[2 * MPF_LT_TEXELS];
    __shared__ uint2 s_box[MPF_LT_MAXS];
    if (pinhole && !aux)
        mpf_wcl_body<HAS_MASK, NL, true, false, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, s_tex, s_box, &src);
    else if (pinhole)
        mpf_wcl_body<HAS_MASK, NL, true, true, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, s_tex, s_box, &src);
    else
        mpf_wcl_body<HAS_MASK, NL, false, true, true>(rgba, quads, params, S, H, W, rgb_out, depth_out, om_out, tgt_mask_out, u8_out, tile, s_tex, s_box, &src);
}

MPF_KNOB g_planar_lds = 1;          // mpf_tune("planar_lds", 0 | 1 | 2): Stage B on the reference's channel-planar tensors by gathers (0), through LDS-staged
                                    // footprints of the workgroup's tile (1) or of every wave's own strip, no barrier in the plane loop (2)

MPF_KNOB g_stage_b_variant = 1;     // mpf_tune("stage_b", v): 0 = v1 reference kernel, 1.. = gather shapes, 20 = LDS-staged footprints

#ifdef MPF_WITNESS
template <bool HAS_MASK>
static int launch_lds(const float *rgba, const MpfViewSet &vs, int V, int S, int H, int W, hipStream_t st)
{
    const unsigned tiles = ((W + 31) / 32) * ((H + 7) / 8);
    hipLaunchKernelGGL((k_warp_composite_lds<HAS_MASK, 2>), dim3(tiles * (unsigned)V), dim3(256), 0, st, rgba, vs, (unsigned)V, S, H, W);
    return mpf_launch_status("k_warp_composite_lds");
}
#endif


template <bool HAS_MASK, int TW, int TH, int WPS>
static int launch_wc2(bool tail_padded, const float *rgba, const float *quads, const float *params, int S, int H, int W, float *rgb,
                      float *depth, float *om, float *tm, uint8_t *u8, hipStream_t st)
{
    const unsigned tiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    dim3 grid(tiles), block(TW * TH);
#define MPF_WC2(NLv, TPv) hipLaunchKernelGGL((k_warp_composite_v2<HAS_MASK, NLv, TW, TH, WPS, TPv>), grid, block, 0, st, rgba, quads, params, S, H, W, rgb, depth, om, tm, u8)
    if (S < 256) { if (tail_padded) MPF_WC2(2, true); else MPF_WC2(2, false); }
    else         { if (tail_padded) MPF_WC2(3, true); else MPF_WC2(3, false); }
#undef MPF_WC2
    return mpf_launch_status("k_warp_composite_v2");
}

template <bool HAS_MASK>
static int dispatch_wc2(int variant, bool tp, const float *rgba, const float *quads, const float *params, int S, int H, int W,
                        float *rgb, float *depth, float *om, float *tm, uint8_t *u8, hipStream_t st)
{
#ifdef MPF_WITNESS
    if (variant >= 101 && variant <= 106) {   // bench-only ablations (invalid results)
        dim3 grid(((W + 63) / 64) * ((H + 3) / 4)), block(256);
        if (variant == 101) hipLaunchKernelGGL((k_warp_composite_dbg<HAS_MASK, 64, 4, 1>), grid, block, 0, st, rgba, quads, params, S, H, W, rgb, depth, om, tm);
        else if (variant == 103) hipLaunchKernelGGL((k_warp_composite_dbg<HAS_MASK, 64, 4, 3>), grid, block, 0, st, rgba, quads, params, S, H, W, rgb, depth, om, tm);
        else if (variant == 105) hipLaunchKernelGGL((k_warp_composite_dbg<HAS_MASK, 64, 4, 5>), grid, block, 0, st, rgba, quads, params, S, H, W, rgb, depth, om, tm);
        else if (variant == 106) hipLaunchKernelGGL((k_warp_composite_dbg<HAS_MASK, 64, 4, 6>), grid, block, 0, st, rgba, quads, params, S, H, W, rgb, depth, om, tm);
        else if (variant == 104) hipLaunchKernelGGL((k_warp_composite_dbg<HAS_MASK, 64, 4, 4>), grid, block, 0, st, rgba, quads, params, S, H, W, rgb, depth, om, tm);
        else hipLaunchKernelGGL((k_warp_composite_dbg<HAS_MASK, 64, 4, 2>), grid, block, 0, st, rgba, quads, params, S, H, W, rgb, depth, om, tm);
        return mpf_launch_status("k_warp_composite_dbg");
    }
    switch (variant) {
    case 2: return launch_wc2<HAS_MASK, 64, 4, 5>(tp, rgba, quads, params, S, H, W, rgb, depth, om, tm, u8, st);
    case 7: return launch_wc2<HAS_MASK, 64, 2, 5>(tp, rgba, quads, params, S, H, W, rgb, depth, om, tm, u8, st);
    case 9: return launch_wc2<HAS_MASK, 64, 4, 4>(tp, rgba, quads, params, S, H, W, rgb, depth, om, tm, u8, st);
    }
#endif
    // 32x8 target tile per workgroup (a wave = 32 px x 2 rows): measured best, the two rows of a wave share a source row
    return launch_wc2<HAS_MASK, 32, 8, 5>(tp, rgba, quads, params, S, H, W, rgb, depth, om, tm, u8, st);
}

template <bool INTERLEAVED, bool HAS_MASK>
static int launch_warp_composite(const float *rgba, const float *quads, const float *params, int S, int H, int W,
                                 float *rgb, float *depth, float *om, float *tm, uint8_t *u8, hipStream_t st)
{
    const unsigned tiles = ((W + MPF_TILE_W - 1) / MPF_TILE_W) * ((H + MPF_TILE_H - 1) / MPF_TILE_H);
    dim3 grid(tiles), block(MPF_TILE_W * MPF_TILE_H);
    if (S < 256)
        hipLaunchKernelGGL((k_warp_composite<INTERLEAVED, HAS_MASK, 2>), grid, block, 0, st, rgba, quads, params, S, H, W,
                           rgb, depth, om, tm, u8);
    else
        hipLaunchKernelGGL((k_warp_composite<INTERLEAVED, HAS_MASK, 3>), grid, block, 0, st, rgba, quads, params, S, H, W,
                           rgb, depth, om, tm, u8);
    return mpf_launch_status("k_warp_composite");
}

static int launch_planar(const MpfPlanarSrc &src, const float *quads, const float *params, int S, int H, int W, float *rgb, float *depth, float *om,
                         float *tm, uint8_t *u8, hipStream_t st)
{
    const unsigned tiles = ((W + 31) / 32) * ((H + 7) / 8);
#ifdef MPF_WITNESS
    if (g_planar_lds == 2 && S <= MPF_WT_MAXS) {
        if (quads) hipLaunchKernelGGL((k_warp_composite_planar_wave<true, 2>), dim3(tiles), dim3(256), 0, st, src, quads, params, S, H, W, rgb, depth, om, tm, u8);
        else hipLaunchKernelGGL((k_warp_composite_planar_wave<false, 2>), dim3(tiles), dim3(256), 0, st, src, quads, params, S, H, W, rgb, depth, om, tm, u8);
        return mpf_launch_status("k_warp_composite_planar_wave");
    }
#endif
    if (g_planar_lds && S <= MPF_LT_MAXS && S < 256) {
        if (quads) hipLaunchKernelGGL((k_warp_composite_planar_lds<true, 2>), dim3(tiles), dim3(256), 0, st, src, quads, params, S, H, W, rgb, depth, om, tm, u8);
        else hipLaunchKernelGGL((k_warp_composite_planar_lds<false, 2>), dim3(tiles), dim3(256), 0, st, src, quads, params, S, H, W, rgb, depth, om, tm, u8);
        return mpf_launch_status("k_warp_composite_planar_lds");
    }
#define MPF_WCP(HM, NLv) hipLaunchKernelGGL((k_warp_composite_planar<HM, NLv>), dim3(tiles), dim3(256), 0, st, src, quads, params, S, H, W, rgb, depth, om, tm, u8)
    if (S < 256) { if (quads) MPF_WCP(true, 2); else MPF_WCP(false, 2); }
    else         { if (quads) MPF_WCP(true, 3); else MPF_WCP(false, 3); }
#undef MPF_WCP
    return mpf_launch_status("k_warp_composite_planar");
}

extern "C" int mpf_warp_composite_split(const float *d_rgb_S3HW, const float *d_sigma_SHW, const float *d_mask_quads, const float *d_params,
                                        int S, int H, int W, float *d_rgb, float *d_depth, float *d_objmask, float *d_tgt_mask,
                                        uint8_t *d_rgb_u8_bgr, void *stream)
{
    MPF_REQUIRE(d_rgb_S3HW && d_sigma_SHW && d_params && d_rgb, "mpf_warp_composite_split: null pointer");
    MPF_REQUIRE(S >= 1 && S < 4096 && H >= 1 && W >= 1, "mpf_warp_composite_split: bad shape S=%d H=%d W=%d", S, H, W);
    MPF_REQUIRE((int64_t)H * W < ((int64_t)1 << 27), "mpf_warp_composite_split: H*W too large for 32-bit byte offsets");
    MPF_REQUIRE((d_mask_quads == nullptr) == (d_objmask == nullptr), "mpf_warp_composite_split: mask quads and objmask output go together");
    MPF_REQUIRE(!d_mask_quads || mpf_aligned16(d_mask_quads), "mpf_warp_composite_split: mask quads must be 16-byte aligned");
    const size_t N4 = (size_t)H * W * 4;
    const MpfPlanarSrc src = { reinterpret_cast<const char *>(d_rgb_S3HW), reinterpret_cast<const char *>(d_sigma_SHW), 3 * N4, N4, (unsigned)(3 * N4), (unsigned)N4 };
    return launch_planar(src, d_mask_quads, d_params, S, H, W, d_rgb, d_depth, d_objmask, d_tgt_mask, d_rgb_u8_bgr, (hipStream_t)stream);
}

extern "C" int mpf_warp_composite(const float *d_rgba, int interleaved, const float *d_mask_quads, const float *d_params,
                                  int S, int H, int W, float *d_rgb, float *d_depth, float *d_objmask,
                                  float *d_tgt_mask, uint8_t *d_rgb_u8_bgr, void *stream)
{
    MPF_REQUIRE(d_rgba && d_params && d_rgb, "mpf_warp_composite: null pointer");
    MPF_REQUIRE(S >= 1 && S < 4096 && H >= 1 && W >= 1, "mpf_warp_composite: bad shape S=%d H=%d W=%d", S, H, W);
    MPF_REQUIRE((int64_t)H * W < ((int64_t)1 << 29), "mpf_warp_composite: H*W too large for 32-bit texel offsets");
    MPF_REQUIRE((d_mask_quads == nullptr) == (d_objmask == nullptr), "mpf_warp_composite: mask quads and objmask output go together");
    MPF_REQUIRE(!interleaved || mpf_aligned16(d_rgba), "mpf_warp_composite: interleaved stack must be 16-byte aligned");
    MPF_REQUIRE(!d_mask_quads || mpf_aligned16(d_mask_quads), "mpf_warp_composite: mask quads must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
#ifdef MPF_WITNESS
    if (interleaved == 2 && g_stage_b_variant == 20 && S <= MPF_LT_MAXS && (int64_t)H * W < ((int64_t)1 << 27)) {
        MpfViewSet vs;
        memset(&vs, 0, sizeof(vs));
        vs.v[0] = MpfWarpView{d_params, d_mask_quads, d_rgb, d_depth, d_objmask, d_tgt_mask, d_rgb_u8_bgr};
        if (d_mask_quads) return launch_lds<true>(d_rgba, vs, 1, S, H, W, st);
        return launch_lds<false>(d_rgba, vs, 1, S, H, W, st);
    }
#endif
    if (interleaved && g_stage_b_variant > 0 && (int64_t)H * W < ((int64_t)1 << 27)) {
        const bool tp = (interleaved == 2);
        if (d_mask_quads) return dispatch_wc2<true>(g_stage_b_variant, tp, d_rgba, d_mask_quads, d_params, S, H, W, d_rgb, d_depth, d_objmask, d_tgt_mask, d_rgb_u8_bgr, st);
        return dispatch_wc2<false>(g_stage_b_variant, tp, d_rgba, nullptr, d_params, S, H, W, d_rgb, d_depth, nullptr, d_tgt_mask, d_rgb_u8_bgr, st);
    }
    if (interleaved) {
        if (d_mask_quads) return launch_warp_composite<true, true>(d_rgba, d_mask_quads, d_params, S, H, W, d_rgb, d_depth, d_objmask, d_tgt_mask, d_rgb_u8_bgr, st);
        return launch_warp_composite<true, false>(d_rgba, nullptr, d_params, S, H, W, d_rgb, d_depth, nullptr, d_tgt_mask, d_rgb_u8_bgr, st);
    }
    if (g_stage_b_variant > 0 && (int64_t)H * W < ((int64_t)1 << 27)) {       // planar stack, fast body
        const size_t N4 = (size_t)H * W * 4;
        const char *base = reinterpret_cast<const char *>(d_rgba);
        const MpfPlanarSrc src = { base, base + 3 * N4, 4 * N4, 4 * N4, (unsigned)(4 * N4), (unsigned)N4 };
        return launch_planar(src, d_mask_quads, d_params, S, H, W, d_rgb, d_depth, d_objmask, d_tgt_mask, d_rgb_u8_bgr, st);
    }
    if (d_mask_quads) return launch_warp_composite<false, true>(d_rgba, d_mask_quads, d_params, S, H, W, d_rgb, d_depth, d_objmask, d_tgt_mask, d_rgb_u8_bgr, st);
    return launch_warp_composite<false, false>(d_rgba, nullptr, d_params, S, H, W, d_rgb, d_depth, nullptr, d_tgt_mask, d_rgb_u8_bgr, st);
}

// mpf_tune("view_shift", k), scheduling only: the odd views of a multi-view launch walk the tile sequence k positions ahead of the even ones.
// With k = 0 the two views of a tile are dispatched back to back, run in lockstep and miss on the same lines at the same moment; a small offset
// (any of +-4 .. +-32, or a whole strip) takes 5-7 % off the two-view launch of bench.py's serial c3 pairs (302-312 -> 288-291 us, the same for
// every sign and size tried: it is the de-synchronisation that helps, not an alignment of the footprints) and 3 % off the pair launch's L2-miss
// traffic at unchanged time; neutral (+-1 %) for the fixed-pose launches of tools/bench_stage_b_views.py (profiles/r3/stage_b_view_shift.log).
MPF_KNOB g_view_shift = 8;

template <bool HAS_MASK>
static int launch_views(bool tp, const float *rgba, const MpfViewSet &vs, int V, int S, int H, int W, hipStream_t st)
{
    constexpr int TW = 32, TH = 8, WPS = 5;
    const unsigned tiles = ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    dim3 grid(tiles * (unsigned)V), block(TW * TH);
#define MPF_WCV(NLv, TPv) hipLaunchKernelGGL((k_warp_composite_views<HAS_MASK, NLv, TW, TH, WPS, TPv>), grid, block, 0, st, rgba, vs, (unsigned)V, S, H, W, (unsigned)g_view_shift % tiles)
    if (S < 256) { if (tp) MPF_WCV(2, true); else MPF_WCV(2, false); }
    else         { if (tp) MPF_WCV(3, true); else MPF_WCV(3, false); }
#undef MPF_WCV
    return mpf_launch_status("k_warp_composite_views");
}

extern "C" int mpf_warp_composite_views(const float *d_rgba, int interleaved, const MpfWarpView *views, int n_views, int S, int H,
                                        int W, void *stream)
{
    MPF_REQUIRE(d_rgba && views, "mpf_warp_composite_views: null pointer");
    MPF_REQUIRE(n_views >= 1 && n_views <= MPF_MAX_VIEWS, "mpf_warp_composite_views: n_views must be 1..%d (got %d)", MPF_MAX_VIEWS, n_views);
    MPF_REQUIRE(interleaved == 1 || interleaved == 2, "mpf_warp_composite_views: the stack must be interleaved [S,H,W,4] (1, or 2 = tail-padded)");
    MPF_REQUIRE(S >= 1 && S < 4096 && H >= 1 && W >= 1, "mpf_warp_composite_views: bad shape S=%d H=%d W=%d", S, H, W);
    MPF_REQUIRE((int64_t)H * W < ((int64_t)1 << 27), "mpf_warp_composite_views: H*W too large for 32-bit byte offsets");
    MPF_REQUIRE(mpf_aligned16(d_rgba), "mpf_warp_composite_views: the stack must be 16-byte aligned");
    MpfViewSet vs;
    memset(&vs, 0, sizeof(vs));
    const bool has_mask = views[0].d_mask_quads != nullptr;
    for (int v = 0; v < n_views; ++v) {
        const MpfWarpView &w = views[v];
        MPF_REQUIRE(w.d_params && w.d_rgb, "mpf_warp_composite_views: view %d: null params / rgb", v);
        MPF_REQUIRE((w.d_mask_quads != nullptr) == has_mask, "mpf_warp_composite_views: all views of a call take a mask, or none does");
        MPF_REQUIRE((w.d_mask_quads == nullptr) == (w.d_objmask == nullptr), "mpf_warp_composite_views: view %d: mask quads and objmask output go together", v);
        MPF_REQUIRE(mpf_aligned16(w.d_mask_quads), "mpf_warp_composite_views: view %d: mask quads must be 16-byte aligned", v);
        vs.v[v] = w;
    }
#ifdef MPF_WITNESS
    if (g_stage_b_variant == 20 && interleaved == 2 && S <= MPF_LT_MAXS) {
        if (has_mask) return launch_lds<true>(d_rgba, vs, n_views, S, H, W, (hipStream_t)stream);
        return launch_lds<false>(d_rgba, vs, n_views, S, H, W, (hipStream_t)stream);
    }
#endif
    if (has_mask) return launch_views<true>(interleaved == 2, d_rgba, vs, n_views, S, H, W, (hipStream_t)stream);
    return launch_views<false>(interleaved == 2, d_rgba, vs, n_views, S, H, W, (hipStream_t)stream);
}

// mask quads ---------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_mask_quads(const float *__restrict__ m, int complement, int H, int W, float4 *__restrict__ q)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int x = (int)(n % W), y = (int)(n / W);
    const bool e = (x + 1) < W, s = (y + 1) < H;
    float a = m[n];
    float b = e ? m[n + 1] : 0.0f;
    float c = s ? m[n + W] : 0.0f;
    float d = (e && s) ? m[n + W + 1] : 0.0f;
    if (complement) {                       // utils/utils.py:225 passes (1 - obj_mask); padding zeros stay zeros
        a = 1.0f - a;
        b = e ? 1.0f - b : 0.0f;
        c = s ? 1.0f - c : 0.0f;
        d = (e && s) ? 1.0f - d : 0.0f;
    }
    q[n] = make_float4(a, b, c, d);
}

extern "C" int mpf_build_mask_quads(const float *d_obj_mask, int complement, int H, int W, float *d_quads, void *stream)
{
    MPF_REQUIRE(d_obj_mask && d_quads && H >= 1 && W >= 1, "mpf_build_mask_quads: bad argument");
    MPF_REQUIRE(mpf_aligned16(d_quads), "mpf_build_mask_quads: output must be 16-byte aligned");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_mask_quads, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_obj_mask,
                       complement, H, W, reinterpret_cast<float4 *>(d_quads));
    return mpf_launch_status("k_mask_quads");
}

// ---------------------------------------------------------------------------------------------------------------
// Stage A + C
// ---------------------------------------------------------------------------------------------------------------

// One thread owns PX pixels (pixel t, t + T, ... with T = number of threads, so every load/store instruction of a wave
// stays a contiguous run) and walks the S planes front to back.
//   A: dist_s = |ray*d_{s+1} - ray*d_s|, T = exp(-sigma*dist), Tacc (double cumprod), blend rgb   (utils/utils.py:190-204)
//   C: w = Tacc*(1-T); flow_p += w * (H_tgt_src[p][s].(x,y,1) / z - (x,y))                       (mpi_rendering.py:102-139,
//                                                                                                 homography_sampler.py:208-218)
// PX = 2 halves the workgroup count so that all of them are resident at once on 256 CUs at 640x960 (2400 workgroups of
// one-pixel threads are 1.17 residency rounds: the second round runs at a fraction of the bandwidth).
#ifndef MPF_NT_STORE
#define MPF_NT_STORE 1   // stream the 629 MB blended stack past the caches: Stage B re-reads it from HBM anyway, and dirty lines left in L2/MALL only delay it
#endif
// ACT: d_mpi holds the RAW last-layer output of the AdaMPI decoder and cum_mask [S,H,W] its cumulative feature mask; the
// activation epilogue of the network (reference model/CPN/decoder.py:166-173: rgb = sigmoid(x), sigma = relu(x * cum_mask) + 1e-4)
// is applied here in registers, so the producer never writes / re-reads an activated copy of the 629 MB stack.
// BLEND = false: flow-only pass (no rgba / planar / tacc output requested): only the sigma channel is read - 4*S*N bytes
// instead of 16*S*N.  The blended stack depends on the image alone, so a caller rendering several pairs of one image
// (the reference's `repeat` loop) blends once and runs this variant per pair.
// DEPTH: register sets of plane loads in flight (the plane loop is unrolled DEPTH times so that every set is addressed statically).
//        2 for the stand-alone kernel, whose 12 waves per CU keep enough bytes in flight between them; the overlapped pair kernel
//        (k_pair_overlap) gives Stage A+C only about one workgroup per CU and makes each wave carry 4-8 planes instead.
struct MpfSbfArgs {
    const float *mpi, *img, *params;
    float flow_clip;
    float *out_rgba, *out_planar, *out_tacc, *flows;
    int64_t T;                      // number of threads that own pixels (thread t owns pixels t, t + T, ...)
    uint8_t *src_u8;
    const float *obj_mask;
    float4 *quads, *quads_c;
    const float *cum_mask;
    int64_t plane_stride, sigma_off;   // floats: plane s starts at mpi + s * plane_stride, its sigma channel sigma_off further
                                       // ([S,4,H,W]: 4 N and 3 N; a bare sigma tensor [S,H,W] for the flow-only pass: N and 0)
};

template <int PX, int P, int NL, bool ACT, bool BLEND, bool NT_STORE, int DEPTH>
MPF_DEV void mpf_sbf_body(const MpfSbfArgs &a, const int S, const int H, const int W, const int64_t t)
{
    const float *__restrict__ mpi = a.mpi;
    const float *__restrict__ img = a.img;
    // d_params is read-only for the whole launch; read through the constant address space so that the per-plane records stay SCALAR
    // loads whatever the compiler can prove about the stores in between (see MpfConstParams)
    const MpfConstParams params = (MpfConstParams)a.params;
    const float *__restrict__ cum_mask = a.cum_mask;
    const float *__restrict__ obj_mask = a.obj_mask;
    float *__restrict__ out_rgba = a.out_rgba;
    float *__restrict__ out_planar = a.out_planar;
    float *__restrict__ out_tacc = a.out_tacc;
    float *__restrict__ flows = a.flows;
    uint8_t *__restrict__ src_u8 = a.src_u8;
    float4 *__restrict__ quads = a.quads;
    float4 *__restrict__ quads_c = a.quads_c;
    const float flow_clip = a.flow_clip;
    const int64_t T = a.T;
    const int64_t N = (int64_t)H * W;
    if (t >= T) return;
    constexpr int NP = (P > 0) ? P : 1;
    constexpr int RS = MPF_PLANE_RECORD * NP;            // floats between two planes' records

    int64_t n[PX];
    bool live[PX];
    float fx[PX], fy[PX], ray[PX][3], im[PX][3], cur[PX][3];
    double acc[PX];
    MpfCsum<NL> cf[PX][NP][2];
    const float d0 = params[MPF_PARAMS_HEADER + 9];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int64_t ni = t + (int64_t)i * T;
        live[i] = ni < N;
        n[i] = live[i] ? ni : (N - 1);                   // dead slots shadow the last pixel: in-range loads, no stores
        fx[i] = (float)(n[i] % W);
        fy[i] = (float)(n[i] / W);
        ray[i][0] = mpf_row3_xy1(params[0], params[1], params[2], fx[i], fy[i]);      // mpi_rendering.py:234
        ray[i][1] = mpf_row3_xy1(params[3], params[4], params[5], fx[i], fy[i]);
        ray[i][2] = mpf_row3_xy1(params[6], params[7], params[8], fx[i], fy[i]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            im[i][c] = img ? img[c * N + n[i]] : 0.0f;                                // no image: the flow-only pass on a bare sigma tensor
            cur[i][c] = ray[i][c] * d0;                                               // :235-236
        }
        acc[i] = 1.0;
#pragma unroll
        for (int p = 0; p < NP; ++p) { cf[i][p][0].init(); cf[i][p][1].init(); }
        // by-products that need nothing but this pixel: the source frame as uint8 BGR (utils/utils.py:174-177) and the
        // bilinear tap quads of obj_mask / 1 - obj_mask for Stage B (see k_mask_quads)
        if (live[i] && src_u8) {
#pragma unroll
            for (int c = 0; c < 3; ++c) src_u8[3 * n[i] + c] = mpf_to_u8(im[i][2 - c]);
        }
        if (live[i] && obj_mask) {
            const int x = (int)(n[i] % W), y = (int)(n[i] / W);
            const bool e = (x + 1) < W, so = (y + 1) < H;
            const float a0 = obj_mask[n[i]];
            const float b = e ? obj_mask[n[i] + 1] : 0.0f;
            const float c2 = so ? obj_mask[n[i] + W] : 0.0f;
            const float d2 = (e && so) ? obj_mask[n[i] + W + 1] : 0.0f;
            if (quads) quads[n[i]] = make_float4(a0, b, c2, d2);
            if (quads_c) quads_c[n[i]] = make_float4(1.0f - a0, e ? 1.0f - b : 0.0f, so ? 1.0f - c2 : 0.0f, (e && so) ? 1.0f - d2 : 0.0f);
        }
    }

    // Software pipeline: the 4 channel loads of planes s+1 .. s+DEPTH-1 are issued before plane s is processed (DEPTH register
    // sets, the loop unrolled DEPTH times) - the kernel is pure streaming and latency x bandwidth decides.
    float ch[DEPTH][PX][4];
    auto load_plane = [&](int s, float (&chs)[PX][4]) {
        const float *pl = mpi + (int64_t)s * a.plane_stride;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            if (BLEND) {
#pragma unroll
                for (int c = 0; c < 3; ++c) chs[i][c] = pl[c * N + n[i]];
            }
            chs[i][3] = pl[a.sigma_off + n[i]];
            if (ACT) {
                const float cm = cum_mask[(int64_t)s * N + n[i]];
                if (BLEND) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) chs[i][c] = 1.0f / (1.0f + mpf_expf_fast(-chs[i][c]));
                }
                chs[i][3] = fmaxf(chs[i][3] * cm, 0.0f) + 1e-4f;
            }
        }
    };
    auto do_plane = [&](int s, const float (&chs)[PX][4]) {
        const MpfConstParams rec = params + MPF_PARAMS_HEADER + RS * s;
        const bool last = (s + 1 == S);
        const float dn = last ? 0.0f : rec[RS + 9];
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            float nx = ray[i][0] * dn, ny = ray[i][1] * dn, nz = ray[i][2] * dn;
            float dist = last ? 1e3f : mpf_norm3_nr(nx - cur[i][0], ny - cur[i][1], nz - cur[i][2]);
            cur[i][0] = nx; cur[i][1] = ny; cur[i][2] = nz;
            const float sg = chs[i][3];
            float Tr = mpf_expf_fast(-sg * dist);
            float alpha = 1.0f - Tr;
            float tacc = (float)acc[i];
            float w = tacc * alpha;
            acc[i] *= (double)(Tr + 1e-6f);
            float one_m = 1.0f - tacc;
            float o[3] = { 0.0f, 0.0f, 0.0f };
            if (BLEND) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float av = tacc * im[i][c];                // blend_weights * src_imgs          utils/utils.py:202-204
                    float bb = one_m * chs[i][c];              // (1 - blend_weights) * mpi_rgb
                    o[c] = av + bb;
                }
            }
            if (BLEND && live[i]) {
                if (out_rgba) {
                    typedef float mpf_v4f __attribute__((ext_vector_type(4)));
                    const mpf_v4f val = { o[0], o[1], o[2], sg };
                    mpf_v4f *dst = reinterpret_cast<mpf_v4f *>(out_rgba) + ((int64_t)s * N + n[i]);
                    if (NT_STORE) __builtin_nontemporal_store(val, dst);
                    else *dst = val;
                }
                if (out_planar) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) out_planar[((int64_t)s * 3 + c) * N + n[i]] = o[c];
                }
                if (out_tacc) out_tacc[(int64_t)s * N + n[i]] = tacc;
            }
            if (P > 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const MpfConstParams h = rec + MPF_PLANE_RECORD * p;
                    float qx = mpf_row3_xy1(h[0], h[1], h[2], fx[i], fy[i]);
                    float qy = mpf_row3_xy1(h[3], h[4], h[5], fx[i], fy[i]);
                    float qz = mpf_row3_xy1(h[6], h[7], h[8], fx[i], fy[i]);
                    const float rz = mpf_rcp_nr(qz);
                    cf[i][p][0].push(w * (mpf_div_nr(qx, qz, rz) - fx[i]));
                    cf[i][p][1].push(w * (mpf_div_nr(qy, qz, rz) - fy[i]));
                }
            }
        }
        if (P > 0 && ((s + 1) & 15) == 0) {
#pragma unroll
            for (int i = 0; i < PX; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) { cf[i][p][0].fold(s + 1); cf[i][p][1].fold(s + 1); }
        }
    };
#pragma unroll
    for (int k = 0; k < DEPTH - 1; ++k)
        if (k < S) load_plane(k, ch[k]);
    int s = 0;
    for (; s + 2 * DEPTH - 2 < S; s += DEPTH) {                         // steady state: branch-free, every register set addressed statically
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            load_plane(s + k + DEPTH - 1, ch[(k + DEPTH - 1) % DEPTH]);
            do_plane(s + k, ch[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 2 * DEPTH - 2; ++k) {                           // the last < 2 DEPTH - 2 planes (uniform branches)
        if (s + k < S) {
            if (s + k + DEPTH - 1 < S) load_plane(s + k + DEPTH - 1, ch[(k + DEPTH - 1) % DEPTH]);
            do_plane(s + k, ch[k % DEPTH]);
        }
    }
    if (P > 0) {
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float f = cf[i][p][k].final();
                    if (flow_clip > 0.0f) f = fminf(fmaxf(f, -flow_clip), flow_clip);   // utils/utils.py:348
                    if (live[i]) flows[((int64_t)p * 2 + k) * N + n[i]] = f;
                }
    }
}

template <int PX, int P, int NL, bool ACT = false, bool BLEND = true, bool NT_STORE = (MPF_NT_STORE != 0)>
__global__ void __launch_bounds__(256)
k_src_blend_flow(const float *__restrict__ mpi, const float *__restrict__ img, const float *__restrict__ params, int S,
                 int H, int W, float flow_clip, float *__restrict__ out_rgba, float *__restrict__ out_planar,
                 float *__restrict__ out_tacc, float *__restrict__ flows, int64_t T, uint8_t *__restrict__ src_u8,
                 const float *__restrict__ obj_mask, float4 *__restrict__ quads, float4 *__restrict__ quads_c,
                 const float *__restrict__ cum_mask, int64_t plane_stride, int64_t sigma_off)
{
    const MpfSbfArgs a = { mpi, img, params, flow_clip, out_rgba, out_planar, out_tacc, flows, T, src_u8, obj_mask, quads, quads_c, cum_mask, plane_stride, sigma_off };
    mpf_sbf_body<PX, P, NL, ACT, BLEND, NT_STORE, 2>(a, S, H, W, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

template <int PX, int P>
static int launch_sbf(const float *mpi, const float *img, const float *params, int S, int H, int W, float clip,
                      float *rgba, float *planar, float *tacc, float *flows, uint8_t *src_u8, const float *om, float *q0, float *q1,
                      const float *cum_mask, hipStream_t st, int64_t plane_stride = 0, int64_t sigma_off = 0)
{
    const int64_t N = (int64_t)H * W;
    const int64_t T = (N + PX - 1) / PX;
    if (plane_stride == 0) { plane_stride = 4 * N; sigma_off = 3 * N; }          // the [S,4,H,W] stack
    dim3 grid((unsigned)((T + 255) / 256)), block(256);
    const bool blend = rgba || planar || tacc;
#define MPF_SBF_GO(NLv, ACTv, BLv) hipLaunchKernelGGL((k_src_blend_flow<PX, P, NLv, ACTv, BLv>), grid, block, 0, st, mpi, img, params, S, H, W, clip, rgba, \
                                              planar, tacc, flows, T, src_u8, om, reinterpret_cast<float4 *>(q0), reinterpret_cast<float4 *>(q1), cum_mask, plane_stride, sigma_off)
#define MPF_SBF_NL(NLv)                                                                              \
    if (cum_mask) { if (blend) MPF_SBF_GO(NLv, true, true); else MPF_SBF_GO(NLv, true, false); }    \
    else          { if (blend) MPF_SBF_GO(NLv, false, true); else MPF_SBF_GO(NLv, false, false); }
    if (S < 256) { MPF_SBF_NL(2) } else { MPF_SBF_NL(3) }
#undef MPF_SBF_NL
#undef MPF_SBF_GO
    return mpf_launch_status("k_src_blend_flow");
}

static int g_sbf_px = 0;   // 0 = auto; tuning knob for benches (mpf_tune)

extern "C" int mpf_src_blend_flow(const float *d_mpi, const float *d_img, const float *d_params, int P, int S, int H, int W,
                                  float flow_clip, float *d_out_rgba, float *d_out_rgb_planar, float *d_out_tacc,
                                  float *d_flows, uint8_t *d_src_u8_bgr, const float *d_obj_mask, float *d_quads,
                                  float *d_quads_complement, const float *d_cum_mask, void *stream)
{
    MPF_REQUIRE(d_mpi && d_img && d_params, "mpf_src_blend_flow: null pointer");
    MPF_REQUIRE((d_quads == nullptr && d_quads_complement == nullptr) || d_obj_mask, "mpf_src_blend_flow: quads need d_obj_mask");
    MPF_REQUIRE(mpf_aligned16(d_quads) && mpf_aligned16(d_quads_complement), "mpf_src_blend_flow: quads must be 16-byte aligned");
    MPF_REQUIRE(P >= 0 && P <= 2, "mpf_src_blend_flow: P must be 0, 1 or 2 (got %d)", P);
    MPF_REQUIRE((P == 0) == (d_flows == nullptr), "mpf_src_blend_flow: flows output iff P > 0");
    MPF_REQUIRE(S >= 1 && S < 4096 && H >= 1 && W >= 1, "mpf_src_blend_flow: bad shape S=%d H=%d W=%d", S, H, W);
    MPF_REQUIRE(!d_out_rgba || mpf_aligned16(d_out_rgba), "mpf_src_blend_flow: rgba output must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int px = g_sbf_px;
    if (px != 1 && px != 2) {
        // residency heuristic (256 CUs x 8 resident 256-thread workgroups): prefer the split that leaves no thin second round
        const int64_t wg1 = ((int64_t)H * W + 255) / 256;
        px = (wg1 > 2048 && wg1 <= 4096) ? 2 : 1;
    }
#define MPF_SBF(PXv)                                                                                                     \
    switch (P) {                                                                                                         \
    case 0: return launch_sbf<PXv, 0>(d_mpi, d_img, d_params, S, H, W, flow_clip, d_out_rgba, d_out_rgb_planar, d_out_tacc, d_flows, d_src_u8_bgr, (d_quads || d_quads_complement) ? d_obj_mask : nullptr, d_quads, d_quads_complement, d_cum_mask, st); \
    case 1: return launch_sbf<PXv, 1>(d_mpi, d_img, d_params, S, H, W, flow_clip, d_out_rgba, d_out_rgb_planar, d_out_tacc, d_flows, d_src_u8_bgr, (d_quads || d_quads_complement) ? d_obj_mask : nullptr, d_quads, d_quads_complement, d_cum_mask, st); \
    default: return launch_sbf<PXv, 2>(d_mpi, d_img, d_params, S, H, W, flow_clip, d_out_rgba, d_out_rgb_planar, d_out_tacc, d_flows, d_src_u8_bgr, (d_quads || d_quads_complement) ? d_obj_mask : nullptr, d_quads, d_quads_complement, d_cum_mask, st); \
    }
    if (px == 2) { MPF_SBF(2) }
    MPF_SBF(1)
#undef MPF_SBF
}

// Stage D for one pixel (utils/utils.py:237-283): thresholds, layer select, uint8 BGR frame, fill mask, merged flow.  Shared by k_merge and
// by the pair launch's Stage A+C role, which can run it as a per-pixel prologue for an EARLIER pair (k_pair_overlap, `mg`).
MPF_DEV void mpf_merge_pixel(const MpfMergeArgs &a, const int64_t n, const int64_t N)
{
    const float th = a.thresh;
    // every input first, unconditionally (all addresses are valid): written as `cond ? a[n] : b[n]` hipcc branches around the loads and
    // the wave sits out six dependent round trips in a kernel that is nothing but latency
    const float om = a.d_obj_mask[n * (a.obj_mask_stride > 1 ? a.obj_mask_stride : 1)], m = a.d_mask[n], md = a.d_mask_dyn[n];
    const float fx = a.d_flow[n], fy = a.d_flow[N + n], gx = a.d_flow_dyn[n], gy = a.d_flow_dyn[N + n];
    float fr[3], fd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fr[c] = a.d_frame[c * N + n];
        fd[c] = a.d_frame_dyn[c * N + n];
    }
    const bool obj = om >= th;                          // source-frame mask   utils/utils.py:270-271, :277-278
    a.d_flow_mix[2 * n] = obj ? fx : gx;
    a.d_flow_mix[2 * n + 1] = obj ? fy : gy;
    const bool sel = m >= th;                           // target-frame masks  :273-276
#pragma unroll
    for (int c = 0; c < 3; ++c) {                       // BGR order           :240-242
        const uint8_t x = (m < th) ? (uint8_t)255 : mpf_to_u8(fr[2 - c]);
        const uint8_t y = (md < th) ? (uint8_t)255 : mpf_to_u8(fd[2 - c]);
        a.d_frame_mix[3 * n + c] = sel ? x : y;
    }
    const float f = sel ? 1.0f : md;                    // :280-283
    a.d_fill_mask[n] = (f < th) ? 1 : 0;
}

// Stage A+C as the overlapped launch runs it (k_pair_overlap): the arithmetic of mpf_sbf_body<.., BLEND = true> - same IEEE operation
// sequence per plane and pixel, bit-identical outputs (tests/test_hip_parity.py) - restated for a role that gets ~1 workgroup per CU
// inside Stage B's register budget:
//   * only what the pipeline needs: the interleaved blended stack, P flows, the per-pixel by-products; no planar / tacc outputs, so
//     no pointer tests in the plane loop, and the last plane (thickness 1e3) is peeled off instead of tested for
//   * buffer addressing: descriptor in SGPRs, plane / channel as the instruction's scalar offset, the pixel as a loop-invariant
//     VGPR offset - no 64-bit VALU adds, no per-channel plane pointers; a dead lane (past the last pixel) shadows the last pixel and
//     stores the same texels again: no exec-mask branch around the stores either.  Needs 16*S*N < 4 GiB (checked by the launcher).
//   * DEPTH planes of loads in flight per wave (ring of register sets, loop unrolled DEPTH times)
template <int PX, int P, int NL, bool ACT, int DEPTH>
MPF_DEV void mpf_sbf_stream(const MpfSbfArgs &a, const int S, const int H, const int W, const int64_t t, const MpfMergeArgs &mg)
{
    const MpfConstParams params = (MpfConstParams)a.params;
    const int64_t N = (int64_t)H * W;
    if (t >= a.T) return;
    constexpr int NP = (P > 0) ? P : 1;
    constexpr int RS = MPF_PLANE_RECORD * NP;
    typedef float mpf_v4f __attribute__((ext_vector_type(4)));

    int64_t n[PX];
    bool live[PX];
    unsigned voff[PX], vout[PX];
    float fx[PX], fy[PX], ray[PX][3], im[PX][3], cur[PX][3];
    double acc[PX];
    MpfCsum<NL> cf[PX][NP][2];
    const float d0 = params[MPF_PARAMS_HEADER + 9];
    const unsigned plane_bytes = (unsigned)N * 4u;
    const __amdgpu_buffer_rsrc_t rs_mpi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.mpi), 0, (unsigned)S * 4u * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ACT ? a.cum_mask : a.mpi), 0, (unsigned)S * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out_rgba, 0, (unsigned)S * 4u * plane_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int64_t ni = t + (int64_t)i * a.T;
        live[i] = ni < N;
        n[i] = live[i] ? ni : (N - 1);
        voff[i] = (unsigned)n[i] * 4u;
        vout[i] = (unsigned)n[i] * 16u;           // a dead lane shadows the last pixel: it stores that pixel's (identical) texels once more
        fx[i] = (float)(n[i] % W);
        fy[i] = (float)(n[i] / W);
        ray[i][0] = mpf_row3_xy1(params[0], params[1], params[2], fx[i], fy[i]);      // mpi_rendering.py:234
        ray[i][1] = mpf_row3_xy1(params[3], params[4], params[5], fx[i], fy[i]);
        ray[i][2] = mpf_row3_xy1(params[6], params[7], params[8], fx[i], fy[i]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            im[i][c] = a.img[c * N + n[i]];
            cur[i][c] = ray[i][c] * d0;                                               // :235-236
        }
        acc[i] = 1.0;
#pragma unroll
        for (int p = 0; p < NP; ++p) { cf[i][p][0].init(); cf[i][p][1].init(); }
        if (live[i] && a.src_u8) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.src_u8[3 * n[i] + c] = mpf_to_u8(im[i][2 - c]);          // utils/utils.py:174-177
        }
        // Stage D of an EARLIER pair, for this pixel.  Its flows may live in the very buffer this thread writes its own flows to at the end
        // (the renderer's two slots alternate): same thread, same addresses, read before written - no other thread touches them.
        if (live[i] && mg.d_flow_mix) mpf_merge_pixel(mg, n[i], N);
        if (live[i] && a.obj_mask) {
            const int x = (int)(n[i] % W), y = (int)(n[i] / W);
            const bool e = (x + 1) < W, so = (y + 1) < H;
            const float a0 = a.obj_mask[n[i]];
            const float b = e ? a.obj_mask[n[i] + 1] : 0.0f;
            const float c2 = so ? a.obj_mask[n[i] + W] : 0.0f;
            const float d2 = (e && so) ? a.obj_mask[n[i] + W + 1] : 0.0f;
            if (a.quads) a.quads[n[i]] = make_float4(a0, b, c2, d2);
            if (a.quads_c) a.quads_c[n[i]] = make_float4(1.0f - a0, e ? 1.0f - b : 0.0f, so ? 1.0f - c2 : 0.0f, (e && so) ? 1.0f - d2 : 0.0f);
        }
    }

    float ch[DEPTH][PX][4];
    auto load_plane = [&](int s, float (&chs)[PX][4]) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                chs[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_mpi, voff[i], (unsigned)(s * 4 + c) * plane_bytes, 0));
            if (ACT) {                                                                // model/CPN/decoder.py:166-173
                const float cm = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_cm, voff[i], (unsigned)s * plane_bytes, 0));
#pragma unroll
                for (int c = 0; c < 3; ++c) chs[i][c] = 1.0f / (1.0f + mpf_expf_fast(-chs[i][c]));
                chs[i][3] = fmaxf(chs[i][3] * cm, 0.0f) + 1e-4f;
            }
        }
    };
    auto do_plane = [&](int s, const float (&chs)[PX][4], auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const MpfConstParams rec = params + MPF_PARAMS_HEADER + RS * s;
        const float dn = LAST ? 0.0f : rec[RS + 9];
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            float dist = 1e3f;                                                        // mpi_rendering.py:73-78
            if (!LAST) {
                const float nx = ray[i][0] * dn, ny = ray[i][1] * dn, nz = ray[i][2] * dn;
                dist = mpf_norm3_nr(nx - cur[i][0], ny - cur[i][1], nz - cur[i][2]);
                cur[i][0] = nx; cur[i][1] = ny; cur[i][2] = nz;
            }
            const float sg = chs[i][3];
            const float Tr = mpf_expf_fast(-sg * dist);
            const float alpha = 1.0f - Tr;
            const float tacc = (float)acc[i];
            const float w = tacc * alpha;
            acc[i] *= (double)(Tr + 1e-6f);
            const float one_m = 1.0f - tacc;
            float o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float av = tacc * im[i][c];              // blend_weights * src_imgs          utils/utils.py:202-204
                const float bb = one_m * chs[i][c];            // (1 - blend_weights) * mpi_rgb
                o[c] = av + bb;
            }
            const mpf_v4f val = { o[0], o[1], o[2], sg };
            // The plane offset is added to the VGPR offset (one VALU add), NOT passed as the instruction's scalar offset.  With an SGPR
            // soffset hipcc assumes the "VMEM store of more than 64 bits, then VALU write of its data VGPRs" hazard away and puts the next
            // plane's first VALU op - which reuses the first data register - right behind the store; on gfx950 the store then sent THAT
            // op's result for lanes 12-15 of every 16 (found by the every-pixel test at 64 x 640 x 960: red channel of every fourth plane,
            // only under load).  With an immediate soffset the hazard recogniser keeps the wait state.
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mpf_v4u, val), rs_out, vout[i] + (unsigned)s * 4u * plane_bytes, 0, 2 /* nt */);
            if (P > 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const MpfConstParams h = rec + MPF_PLANE_RECORD * p;
                    const float qx = mpf_row3_xy1(h[0], h[1], h[2], fx[i], fy[i]);
                    const float qy = mpf_row3_xy1(h[3], h[4], h[5], fx[i], fy[i]);
                    const float qz = mpf_row3_xy1(h[6], h[7], h[8], fx[i], fy[i]);
                    const float rz = mpf_rcp_nr(qz);
                    cf[i][p][0].push(w * (mpf_div_nr(qx, qz, rz) - fx[i]));
                    cf[i][p][1].push(w * (mpf_div_nr(qy, qz, rz) - fy[i]));
                }
            }
        }
        if (P > 0 && ((s + 1) & 15) == 0) {
#pragma unroll
            for (int i = 0; i < PX; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) { cf[i][p][0].fold(s + 1); cf[i][p][1].fold(s + 1); }
        }
    };
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;
#pragma unroll
    for (int k = 0; k < DEPTH - 1; ++k)
        if (k < S) load_plane(k, ch[k]);
    int s = 0;
    for (; s + 2 * DEPTH - 2 < S; s += DEPTH) {                         // steady state: branch-free; none of these planes is the last
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            load_plane(s + k + DEPTH - 1, ch[(k + DEPTH - 1) % DEPTH]);
            do_plane(s + k, ch[k], No());
        }
    }
#pragma unroll
    for (int k = 0; k < 2 * DEPTH - 2; ++k) {                           // the last < 2 DEPTH - 2 planes (uniform branches)
        if (s + k < S) {
            if (s + k + DEPTH - 1 < S) load_plane(s + k + DEPTH - 1, ch[(k + DEPTH - 1) % DEPTH]);
            if (s + k + 1 == S) do_plane(s + k, ch[k % DEPTH], Yes());
            else do_plane(s + k, ch[k % DEPTH], No());
        }
    }
    if (P > 0) {
#pragma unroll
        for (int i = 0; i < PX; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float f = cf[i][p][k].final();
                    if (a.flow_clip > 0.0f) f = fminf(fmaxf(f, -a.flow_clip), a.flow_clip);   // utils/utils.py:348
                    if (live[i]) a.flows[((int64_t)p * 2 + k) * N + n[i]] = f;
                }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage B of image i  ||  Stage A+C of image i+1, in ONE launch (the reference's unit of work, utils/utils.py:190-236, is one
// source-frame pass plus two target-frame passes; back to back they are an HBM-bound kernel followed by a VALU-issue-bound one).
//
// Run as two kernels on two streams they barely overlap: each is sized to fill the chip on its own (Stage B takes 480 of the 512
// VGPRs of every SIMD), so the second only gets the first one's tail.  Here the GRID is heterogeneous instead: the k-th workgroup
// dispatched to an XCD is a Stage A+C workgroup for KA out of every KA + KB positions (a Bresenham pattern) and a Stage B workgroup
// otherwise, both compiled into one kernel with Stage B's register budget (5 waves per SIMD).  Workgroups are dispatched in
// index order as slots free up, so both kinds advance through their lists at the same pace and finish together, and every CU
// holds a mix of them at any time: while the A+C waves wait on HBM (they stream 1.27 GB) the Stage B waves issue arithmetic.
// Both roles run exactly the bodies of the stand-alone kernels (mpf_wc2_select, mpf_sbf_body), so results are bit-identical.
//   * Stage B keeps its XCD-aware strip order: its logical index is recovered from (xcd, rank among the B blocks of that XCD).
//   * Stage A+C gets only ~1 workgroup per CU, so each wave carries DEPTH (4-8) planes of loads in flight instead of 2, one pixel
//     per thread (46-62 VGPRs, inside Stage B's 96).
// ---------------------------------------------------------------------------------------------------------------
#ifndef MPF_OVL_PX
#define MPF_OVL_PX 1            // pixels per thread of the overlapped Stage A+C role
#endif
template <bool HAS_MASK, int NL, int P, bool ACT, int DEPTH>
__global__ void __launch_bounds__(256, 5)
k_pair_overlap(const float *__restrict__ rgba_b, const MpfViewSet vs, const unsigned V, const MpfSbfArgs ac, const int S, const int H, const int W,
               const unsigned nB, const unsigned nA, const unsigned KB, const unsigned KA, const int ablate_, const unsigned view_shift, const unsigned xcd_a_,
               const MpfMergeArgs mg)
{
    constexpr int TW = 32, TH = 8;
    const unsigned xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;          // the k-th workgroup of this XCD
    bool role_a;
    unsigned ja = 0, jb = 0, l = 0;
    const int ablate = MPF_WIT(ablate_);                                // (the product build compiles neither the ablations nor the by-XCD role split)
    const unsigned xcd_a = MPF_WIT(xcd_a_);
    if (xcd_a) {
        // roles partitioned by XCD (mpf_tune("ovl_xcd_a", n)): the first n XCDs run Stage A+C workgroups only, the others Stage B only, so
        // the streaming role's 1.27 GB cannot evict the gather role's texel rows from the L2 they are re-used in
        role_a = xcd < xcd_a;
        ja = k * xcd_a + xcd;
        jb = k * (8u - xcd_a) + (xcd - xcd_a);
        if (!role_a) l = mpf_xcd_remap_n(jb, nB, 8u - xcd_a);
    } else {
        const unsigned per = KB + KA;
        // 64-bit: k * KA passes 2^32 for shallow stacks of very large frames (S <= 7 with 16 views at ~2^25 px)
        const unsigned a0 = (unsigned)(((uint64_t)k * KA) / per), a1 = (unsigned)(((uint64_t)(k + 1u) * KA) / per);   // A+C workgroups among the first k / k + 1 of this XCD
        role_a = a1 > a0;
        ja = a0 * 8u + xcd;
        jb = (k - a0) * 8u + xcd;
        if (!role_a) l = mpf_xcd_remap(jb, nB);
    }
    if (role_a) {
        if (ja >= nA || (ablate & 3) == 1) return;                       // ablate (bench only): 1 = Stage B workgroups only, 2 = Stage A+C only
        if (((ablate >> 2) & 3) == 1) __builtin_amdgcn_s_setprio(1);      // bits 2-3: wave priority of the A+C role (tuning experiment)
        else if (((ablate >> 2) & 3) == 2) __builtin_amdgcn_s_setprio(2);
        else if (((ablate >> 2) & 3) == 3) __builtin_amdgcn_s_setprio(3);
        mpf_sbf_stream<MPF_OVL_PX, P, NL, ACT, DEPTH>(ac, S, H, W, (int64_t)ja * 256 + threadIdx.x, mg);
    } else {
        if (jb >= nB || (ablate & 3) == 2) return;
        if (((ablate >> 4) & 3) == 1) __builtin_amdgcn_s_setprio(1);      // bits 4-5: wave priority of the Stage B role
        else if (((ablate >> 4) & 3) == 3) __builtin_amdgcn_s_setprio(3);
        const unsigned view = l % V;
        const unsigned tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, ntiles = tiles_x * tiles_y;
        const unsigned seq = (view & 1u) ? (l / V + view_shift) % ntiles : l / V;        // see k_warp_composite_views
        const unsigned tile = mpf_strip_order(seq, tiles_x, tiles_y);
        const MpfWarpView &w = vs.v[view];
        mpf_wc2_select<HAS_MASK, NL, TW, TH, true>(rgba_b, w.d_mask_quads, w.d_params, S, H, W, w.d_rgb, w.d_depth, w.d_objmask, w.d_tgt_mask,
                                                   w.d_rgb_u8_bgr, tile);
    }
}

MPF_KNOB g_ovl_depth = 4;       // mpf_tune("ovl_depth", 4 | 8): no measurable difference at 64x640x960 (both 492-523 us per launch on one box)
MPF_KNOB g_ovl_ablate = 0;      // mpf_tune("ovl_ablate", 0 | 1 | 2): bench-only, results invalid when non-zero
MPF_KNOB g_ovl_xcd_a = 0;       // mpf_tune("ovl_xcd_a", 0..7): 0 = both roles interleaved on every XCD (Bresenham), n = the first n XCDs run Stage A+C only

template <bool HAS_MASK, int NL, int P, bool ACT>
static int launch_overlap(const float *rgba_b, const MpfViewSet &vs, unsigned V, const MpfSbfArgs &ac, int S, int H, int W, hipStream_t st, const MpfMergeArgs &mg)
{
    const unsigned tiles = ((W + 31) / 32) * ((H + 7) / 8);
    const unsigned nB = tiles * V;
    const unsigned nA = (unsigned)((ac.T + 255) / 256);              // ac.T = threads of the A+C role (N / MPF_OVL_PX, rounded up)
    const unsigned KB = (nB + 7) / 8, KA = (nA + 7) / 8;
    const unsigned xa = (unsigned)g_ovl_xcd_a;
    unsigned per_xcd = KB + KA;
    if (xa) {                                                        // roles by XCD: every XCD gets as many workgroups as the busiest one needs (the rest exit at once)
        const unsigned pa = (nA + xa - 1) / xa, pb = (nB + (8u - xa) - 1) / (8u - xa);
        per_xcd = pa > pb ? pa : pb;
    }
    dim3 grid(8u * per_xcd), block(256);
#ifdef MPF_WITNESS
    if (g_ovl_depth == 8) {
        hipLaunchKernelGGL((k_pair_overlap<HAS_MASK, NL, P, ACT, 8>), grid, block, 0, st, rgba_b, vs, V, ac, S, H, W, nB, nA, KB, KA, g_ovl_ablate, (unsigned)g_view_shift % tiles, xa, mg);
        return mpf_launch_status("k_pair_overlap");
    }
#endif
    hipLaunchKernelGGL((k_pair_overlap<HAS_MASK, NL, P, ACT, 4>), grid, block, 0, st, rgba_b, vs, V, ac, S, H, W, nB, nA, KB, KA, g_ovl_ablate, (unsigned)g_view_shift % tiles, xa, mg);
    return mpf_launch_status("k_pair_overlap");
}

extern "C" int mpf_warp_views_and_blend_next(const float *d_rgba, const MpfWarpView *views, int n_views,
                                             const float *d_mpi_next, const float *d_img_next, const float *d_params_next, int P,
                                             float flow_clip, float *d_out_rgba_next, float *d_flows_next, uint8_t *d_src_u8_bgr_next,
                                             const float *d_obj_mask_next, float *d_quads_next, float *d_quads_complement_next,
                                             const float *d_cum_mask_next, int S, int H, int W, void *stream)
{
    return mpf_warp_views_blend_next_merge_prev(d_rgba, views, n_views, d_mpi_next, d_img_next, d_params_next, P, flow_clip, d_out_rgba_next, d_flows_next,
                                                d_src_u8_bgr_next, d_obj_mask_next, d_quads_next, d_quads_complement_next, d_cum_mask_next, S, H, W, nullptr, stream);
}

extern "C" int mpf_warp_views_blend_next_merge_prev(const float *d_rgba, const MpfWarpView *views, int n_views,
                                                    const float *d_mpi_next, const float *d_img_next, const float *d_params_next, int P,
                                                    float flow_clip, float *d_out_rgba_next, float *d_flows_next, uint8_t *d_src_u8_bgr_next,
                                                    const float *d_obj_mask_next, float *d_quads_next, float *d_quads_complement_next,
                                                    const float *d_cum_mask_next, int S, int H, int W, const MpfMergeArgs *merge_prev, void *stream)
{
    MpfMergeArgs mg;
    memset(&mg, 0, sizeof(mg));
    if (merge_prev) {
        mg = *merge_prev;
        MPF_REQUIRE(mg.d_frame && mg.d_frame_dyn && mg.d_mask && mg.d_mask_dyn && mg.d_flow && mg.d_flow_dyn && mg.d_obj_mask && mg.d_flow_mix && mg.d_frame_mix &&
                        mg.d_fill_mask, "mpf_warp_views_blend_next_merge_prev: merge_prev has a null pointer");
        for (int v = 0; v < n_views && views; ++v)
            MPF_REQUIRE(views[v].d_rgb != mg.d_frame && views[v].d_rgb != mg.d_frame_dyn && views[v].d_objmask != mg.d_mask && views[v].d_objmask != mg.d_mask_dyn,
                        "mpf_warp_views_blend_next_merge_prev: the merged pair's views must not be the views this launch renders");
        MPF_REQUIRE(mg.obj_mask_stride >= 0 && mg.obj_mask_stride <= 4, "mpf_warp_views_blend_next_merge_prev: obj_mask_stride must be 0..4");
        // The folded merge is race-free only if (1) the thread that merges pixel n is the ONLY one that writes anything the merge of pixel n reads:
        // the merged pair's flows are either disjoint from the flows this launch writes or exactly those planes (d_flows_next + {0, 2N}: same pixel,
        // same thread, read before written), its object mask is disjoint from the launch's outputs or exactly the .x of d_quads_next (same argument),
        // and (2) what the merge writes overlaps nothing this launch reads or writes.
        const size_t Nn = (size_t)H * W;
        auto overlaps = [](const void *a, size_t an, const void *b, size_t bn) {
            return a && b && (const char *)a < (const char *)b + bn && (const char *)b < (const char *)a + an;
        };
        const size_t flows_next_bytes = (size_t)P * 2 * Nn * sizeof(float);
        const float *fl[2] = { mg.d_flow, mg.d_flow_dyn };
        for (int k = 0; k < 2; ++k) {
            const bool aligned = d_flows_next && P == 2 && (fl[k] == d_flows_next || fl[k] == d_flows_next + 2 * Nn);
            MPF_REQUIRE(aligned || !overlaps(fl[k], 2 * Nn * sizeof(float), d_flows_next, flows_next_bytes),
                        "mpf_warp_views_blend_next_merge_prev: merge_prev's flows must be disjoint from d_flows_next or exactly its two pose planes");
        }
        const size_t om_bytes = Nn * sizeof(float) * (size_t)(mg.obj_mask_stride > 1 ? mg.obj_mask_stride : 1);
        const bool om_is_quads = mg.obj_mask_stride == 4 && (const void *)mg.d_obj_mask == (const void *)d_quads_next;
        MPF_REQUIRE(om_is_quads || (!overlaps(mg.d_obj_mask, om_bytes, d_quads_next, Nn * 16) && !overlaps(mg.d_obj_mask, om_bytes, d_quads_complement_next, Nn * 16) &&
                                    !overlaps(mg.d_obj_mask, om_bytes, d_out_rgba_next, (size_t)S * Nn * 16) && !overlaps(mg.d_obj_mask, om_bytes, d_flows_next, flows_next_bytes)),
                    "mpf_warp_views_blend_next_merge_prev: merge_prev's object mask must not be a buffer this launch writes (other than the .x of d_quads_next, stride 4)");
        const struct { const void *p; size_t n; const char *what; } outs[3] = { { mg.d_flow_mix, 2 * Nn * sizeof(float), "flow_mix" }, { mg.d_frame_mix, 3 * Nn, "frame_mix" },
                                                                                  { mg.d_fill_mask, Nn, "fill_mask" } };
        for (int k = 0; k < 3; ++k) {
            bool bad = overlaps(outs[k].p, outs[k].n, d_out_rgba_next, (size_t)S * Nn * 16) || overlaps(outs[k].p, outs[k].n, d_rgba, (size_t)S * Nn * 16) ||
                       overlaps(outs[k].p, outs[k].n, d_flows_next, flows_next_bytes) || overlaps(outs[k].p, outs[k].n, d_quads_next, Nn * 16) ||
                       overlaps(outs[k].p, outs[k].n, d_quads_complement_next, Nn * 16) || overlaps(outs[k].p, outs[k].n, d_src_u8_bgr_next, 3 * Nn) ||
                       overlaps(outs[k].p, outs[k].n, d_obj_mask_next, Nn * 4) || overlaps(outs[k].p, outs[k].n, mg.d_obj_mask, om_bytes) ||
                       overlaps(outs[k].p, outs[k].n, mg.d_flow, 2 * Nn * 4) || overlaps(outs[k].p, outs[k].n, mg.d_flow_dyn, 2 * Nn * 4) ||
                       overlaps(outs[k].p, outs[k].n, mg.d_frame, 3 * Nn * 4) || overlaps(outs[k].p, outs[k].n, mg.d_frame_dyn, 3 * Nn * 4) ||
                       overlaps(outs[k].p, outs[k].n, mg.d_mask, Nn * 4) || overlaps(outs[k].p, outs[k].n, mg.d_mask_dyn, Nn * 4);
            for (int j = k + 1; j < 3; ++j) bad = bad || overlaps(outs[k].p, outs[k].n, outs[j].p, outs[j].n);
            for (int v = 0; v < n_views && views; ++v)
                bad = bad || overlaps(outs[k].p, outs[k].n, views[v].d_rgb, 3 * Nn * 4) || overlaps(outs[k].p, outs[k].n, views[v].d_objmask, Nn * 4) ||
                      overlaps(outs[k].p, outs[k].n, views[v].d_depth, Nn * 4) || overlaps(outs[k].p, outs[k].n, views[v].d_tgt_mask, Nn * 4) ||
                      overlaps(outs[k].p, outs[k].n, views[v].d_rgb_u8_bgr, 3 * Nn) || overlaps(outs[k].p, outs[k].n, views[v].d_mask_quads, Nn * 16);
            MPF_REQUIRE(!bad, "mpf_warp_views_blend_next_merge_prev: merge_prev's %s overlaps a buffer this launch reads or writes", outs[k].what);
        }
    }
    MPF_REQUIRE(d_rgba && views && d_mpi_next && d_img_next && d_params_next && d_out_rgba_next, "mpf_warp_views_and_blend_next: null pointer");
    MPF_REQUIRE(d_out_rgba_next != d_rgba, "mpf_warp_views_and_blend_next: the stack being rendered and the stack being written must be different buffers");
    MPF_REQUIRE(n_views >= 1 && n_views <= MPF_MAX_VIEWS, "mpf_warp_views_and_blend_next: n_views must be 1..%d (got %d)", MPF_MAX_VIEWS, n_views);
    MPF_REQUIRE(S >= 1 && S < 4096 && H >= 1 && W >= 1, "mpf_warp_views_and_blend_next: bad shape S=%d H=%d W=%d", S, H, W);
    MPF_REQUIRE((int64_t)H * W < ((int64_t)1 << 27), "mpf_warp_views_and_blend_next: H*W too large for 32-bit byte offsets");
    MPF_REQUIRE((int64_t)S * H * W * 16 < ((int64_t)1 << 32), "mpf_warp_views_and_blend_next: the plane stack must be smaller than 4 GiB (buffer addressing); use the two separate calls");
    MPF_REQUIRE(mpf_aligned16(d_rgba) && mpf_aligned16(d_out_rgba_next), "mpf_warp_views_and_blend_next: the stacks must be 16-byte aligned");
    MPF_REQUIRE(P >= 0 && P <= 2 && (P == 0) == (d_flows_next == nullptr), "mpf_warp_views_and_blend_next: P must be 0..2, flows output iff P > 0");
    MPF_REQUIRE((d_quads_next == nullptr && d_quads_complement_next == nullptr) || d_obj_mask_next, "mpf_warp_views_and_blend_next: quads need d_obj_mask_next");
    MPF_REQUIRE(mpf_aligned16(d_quads_next) && mpf_aligned16(d_quads_complement_next), "mpf_warp_views_and_blend_next: quads must be 16-byte aligned");
    MpfViewSet vs;
    memset(&vs, 0, sizeof(vs));
    const bool has_mask = views[0].d_mask_quads != nullptr;
    for (int v = 0; v < n_views; ++v) {
        const MpfWarpView &w = views[v];
        MPF_REQUIRE(w.d_params && w.d_rgb, "mpf_warp_views_and_blend_next: view %d: null params / rgb", v);
        MPF_REQUIRE((w.d_mask_quads != nullptr) == has_mask, "mpf_warp_views_and_blend_next: all views of a call take a mask, or none does");
        MPF_REQUIRE((w.d_mask_quads == nullptr) == (w.d_objmask == nullptr), "mpf_warp_views_and_blend_next: view %d: mask quads and objmask output go together", v);
        MPF_REQUIRE(mpf_aligned16(w.d_mask_quads), "mpf_warp_views_and_blend_next: view %d: mask quads must be 16-byte aligned", v);
        vs.v[v] = w;
    }
    const int64_t N = (int64_t)H * W;
    const MpfSbfArgs ac = { d_mpi_next, d_img_next, d_params_next, flow_clip, d_out_rgba_next, nullptr, nullptr, d_flows_next, (N + MPF_OVL_PX - 1) / MPF_OVL_PX, d_src_u8_bgr_next,
                            (d_quads_next || d_quads_complement_next) ? d_obj_mask_next : nullptr, reinterpret_cast<float4 *>(d_quads_next),
                            reinterpret_cast<float4 *>(d_quads_complement_next), d_cum_mask_next, 4 * N, 3 * N };
    hipStream_t st = (hipStream_t)stream;
#define MPF_OVL(HM, NLv)                                                                                                        \
    switch (P) {                                                                                                                \
    case 0: return d_cum_mask_next ? launch_overlap<HM, NLv, 0, true>(d_rgba, vs, n_views, ac, S, H, W, st, mg) : launch_overlap<HM, NLv, 0, false>(d_rgba, vs, n_views, ac, S, H, W, st, mg); \
    case 1: return d_cum_mask_next ? launch_overlap<HM, NLv, 1, true>(d_rgba, vs, n_views, ac, S, H, W, st, mg) : launch_overlap<HM, NLv, 1, false>(d_rgba, vs, n_views, ac, S, H, W, st, mg); \
    default: return d_cum_mask_next ? launch_overlap<HM, NLv, 2, true>(d_rgba, vs, n_views, ac, S, H, W, st, mg) : launch_overlap<HM, NLv, 2, false>(d_rgba, vs, n_views, ac, S, H, W, st, mg); \
    }
    if (S < 256) { if (has_mask) { MPF_OVL(true, 2) } else { MPF_OVL(false, 2) } }
    else         { if (has_mask) { MPF_OVL(true, 3) } else { MPF_OVL(false, 3) } }
#undef MPF_OVL
}

// hard_flow = True (utils/mpi/mpi_rendering.py:126-130): the flow of the arg-max-weight plane instead of the weighted sum.  One pass over the
// sigma planes: the weights w_s = Tacc_s (1 - T_s) with the IEEE sequence of mpf_sbf_body (so the arg max is the reference's), the per-plane flow of
// every pose in registers, the first maximal weight wins (torch.argmax returns the first maximal index).  Nothing per-plane is materialised - the
// generic path (mpf_homography_flow + mpf_volume_render(hard)) writes and re-reads S x 2 x H x W flows per pose.
template <int P>
__global__ void __launch_bounds__(256)
k_src_flow_hard(const float *__restrict__ sigma, const int64_t plane_stride, const float *__restrict__ params_global, int S, int H, int W, float flow_clip,
                float *__restrict__ flows)
{
    const MpfConstParams params = (MpfConstParams)params_global;
    constexpr int RS = MPF_PLANE_RECORD * P;
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float fx = (float)(n % W), fy = (float)(n / W);
    float ray[3], cur[3];
    ray[0] = mpf_row3_xy1(params[0], params[1], params[2], fx, fy);               // mpi_rendering.py:234
    ray[1] = mpf_row3_xy1(params[3], params[4], params[5], fx, fy);
    ray[2] = mpf_row3_xy1(params[6], params[7], params[8], fx, fy);
    const float d0 = params[MPF_PARAMS_HEADER + 9];
#pragma unroll
    for (int c = 0; c < 3; ++c) cur[c] = ray[c] * d0;                             // :235-236
    double acc = 1.0;
    float best = -INFINITY, bf[P][2];
#pragma unroll
    for (int p = 0; p < P; ++p) bf[p][0] = bf[p][1] = 0.0f;
    constexpr int D = 4;                                                          // sigma loads in flight
    for (int s0 = 0; s0 < S; s0 += D) {
        float sg[D];
#pragma unroll
        for (int k = 0; k < D; ++k) sg[k] = sigma[(int64_t)min(s0 + k, S - 1) * plane_stride + n];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int s = s0 + k;
            if (s >= S) break;
            const MpfConstParams rec = params + MPF_PARAMS_HEADER + RS * s;
            const bool last = (s + 1 == S);
            const float dn = last ? 0.0f : rec[RS + 9];
            const float nx = ray[0] * dn, ny = ray[1] * dn, nz = ray[2] * dn;
            const float dist = last ? 1e3f : mpf_norm3_nr(nx - cur[0], ny - cur[1], nz - cur[2]);
            cur[0] = nx; cur[1] = ny; cur[2] = nz;
            const float Tr = mpf_expf_fast(-sg[k] * dist);
            const float alpha = 1.0f - Tr;
            const float tacc = (float)acc;
            const float w = tacc * alpha;
            acc *= (double)(Tr + 1e-6f);
            if (w > best) {                                                        // the FIRST maximal weight
                best = w;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const MpfConstParams h = rec + MPF_PLANE_RECORD * p;
                    const float qx = mpf_row3_xy1(h[0], h[1], h[2], fx, fy);
                    const float qy = mpf_row3_xy1(h[3], h[4], h[5], fx, fy);
                    const float qz = mpf_row3_xy1(h[6], h[7], h[8], fx, fy);
                    const float rz = mpf_rcp_nr(qz);
                    bf[p][0] = mpf_div_nr(qx, qz, rz) - fx;
                    bf[p][1] = mpf_div_nr(qy, qz, rz) - fy;
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float f = bf[p][k];
            if (flow_clip > 0.0f) f = fminf(fmaxf(f, -flow_clip), flow_clip);       // utils/utils.py:348
            flows[((int64_t)p * 2 + k) * N + n] = f;
        }
}

extern "C" int mpf_src_flow_hard(const float *d_sigma, int64_t plane_stride, const float *d_params, int P, int S, int H, int W, float flow_clip, float *d_flows,
                                 void *stream)
{
    MPF_REQUIRE(d_sigma && d_params && d_flows, "mpf_src_flow_hard: null pointer");
    MPF_REQUIRE(P >= 1 && P <= 2, "mpf_src_flow_hard: P must be 1 or 2 (got %d)", P);
    MPF_REQUIRE(S >= 1 && S < 4096 && H >= 1 && W >= 1, "mpf_src_flow_hard: bad shape S=%d H=%d W=%d", S, H, W);
    const int64_t N = (int64_t)H * W;
    MPF_REQUIRE(plane_stride >= N, "mpf_src_flow_hard: plane_stride (floats between consecutive sigma planes) must be at least H*W");
    const dim3 grid((unsigned)((N + 255) / 256));
    if (P == 1) hipLaunchKernelGGL((k_src_flow_hard<1>), grid, dim3(256), 0, (hipStream_t)stream, d_sigma, plane_stride, d_params, S, H, W, flow_clip, d_flows);
    else hipLaunchKernelGGL((k_src_flow_hard<2>), grid, dim3(256), 0, (hipStream_t)stream, d_sigma, plane_stride, d_params, S, H, W, flow_clip, d_flows);
    return mpf_launch_status("k_src_flow_hard");
}

extern "C" int mpf_src_flow(const float *d_sigma_SHW, const float *d_params, int P, int S, int H, int W, float flow_clip, float *d_flows, void *stream)
{
    MPF_REQUIRE(d_sigma_SHW && d_params && d_flows, "mpf_src_flow: null pointer");
    MPF_REQUIRE(P >= 1 && P <= 2, "mpf_src_flow: P must be 1 or 2 (got %d)", P);
    MPF_REQUIRE(S >= 1 && S < 4096 && H >= 1 && W >= 1, "mpf_src_flow: bad shape S=%d H=%d W=%d", S, H, W);
    const int64_t N = (int64_t)H * W;
    hipStream_t st = (hipStream_t)stream;
    // flow-only body (BLEND = false): reads nothing but the sigma planes; no image, no by-products
    if (P == 1) return launch_sbf<1, 1>(d_sigma_SHW, nullptr, d_params, S, H, W, flow_clip, nullptr, nullptr, nullptr, d_flows, nullptr, nullptr, nullptr, nullptr, nullptr, st, N, 0);
    return launch_sbf<1, 2>(d_sigma_SHW, nullptr, d_params, S, H, W, flow_clip, nullptr, nullptr, nullptr, d_flows, nullptr, nullptr, nullptr, nullptr, nullptr, st, N, 0);
}

#ifdef MPF_WITNESS
void mpf_fwarp_set_path(int v);      // mpf_fwarp.hip
#endif
void mpf_fwarp_set_gate(int v);
void mpf_conv_set_prefetch(int v);   // mpf_conv.hip
void mpf_fwarp_set_prio(int v);
void mpf_fwarp_set_grid(int v);

extern "C" int mpf_tune(const char *key, int value)
{
    // the product's knobs: scheduling only - launch shapes, grid caps, wave priorities, prefetch, the gather / radix gate - every setting gives the same bytes
    if (key && !strcmp(key, "sbf_px")) { g_sbf_px = value; return 0; }
    if (key && !strcmp(key, "fwarp_gate")) { mpf_fwarp_set_gate(value); return 0; }
    if (key && !strcmp(key, "conv_pf")) { mpf_conv_set_prefetch(value); return 0; }
    if (key && !strcmp(key, "chain_grid")) { mpf_fwarp_set_grid(value); return 0; }
    if (key && !strcmp(key, "chain_prio")) { mpf_fwarp_set_prio(value); return 0; }
#ifdef MPF_WITNESS
    // retired kernel variants (bit-identical witnesses) and timing ablations (INVALID results): libmpiflow_hip_witness.so only
    if (key && !strcmp(key, "stage_b")) { g_stage_b_variant = value; return 0; }
    if (key && !strcmp(key, "planar_lds")) { g_planar_lds = (value < 0 || value > 2) ? 1 : value; return 0; }
    if (key && !strcmp(key, "ovl_depth")) { g_ovl_depth = (value == 8) ? 8 : 4; return 0; }
    if (key && !strcmp(key, "ovl_ablate")) { g_ovl_ablate = value; return 0; }
    if (key && !strcmp(key, "view_shift")) { g_view_shift = value < 0 ? 0 : value; return 0; }
    if (key && !strcmp(key, "fwarp_path")) { mpf_fwarp_set_path(value); return 0; }
    if (key && !strcmp(key, "ovl_xcd_a")) { g_ovl_xcd_a = (value < 0 || value > 7) ? 0 : value; return 0; }
    mpf_set_error("mpf_tune: unknown key");
#else
    mpf_set_error("mpf_tune: unknown key (variant / ablation keys exist in the witness build only: libmpiflow_hip_witness.so)");
#endif
    return MPF_ERR_BAD_ARGUMENT;
}

extern "C" int mpf_is_witness_build(void)
{
#ifdef MPF_WITNESS
    return 1;
#else
    return 0;
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Stage D and uint8 conversion
// ---------------------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_merge(const MpfMergeArgs m, int64_t N)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    mpf_merge_pixel(m, n, N);
}

extern "C" int mpf_merge(const float *d_frame, const float *d_frame_dyn, const float *d_mask, const float *d_mask_dyn,
                         const float *d_flow, const float *d_flow_dyn, const float *d_obj_mask, float thresh, int H, int W,
                         float *d_flow_mix, uint8_t *d_frame_mix, uint8_t *d_fill_mask, void *stream)
{
    MPF_REQUIRE(d_frame && d_frame_dyn && d_mask && d_mask_dyn && d_flow && d_flow_dyn && d_obj_mask && d_flow_mix &&
                    d_frame_mix && d_fill_mask && H >= 1 && W >= 1, "mpf_merge: bad argument");
    const int64_t N = (int64_t)H * W;
    const MpfMergeArgs m = { d_frame, d_frame_dyn, d_mask, d_mask_dyn, d_flow, d_flow_dyn, d_obj_mask, thresh, d_flow_mix, d_frame_mix, d_fill_mask, 1 };
    hipLaunchKernelGGL(k_merge, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m, N);
    return mpf_launch_status("k_merge");
}

extern "C" int mpf_merge_ex(const MpfMergeArgs *args, int H, int W, void *stream)
{
    MPF_REQUIRE(args && H >= 1 && W >= 1, "mpf_merge_ex: bad argument");
    const MpfMergeArgs &m = *args;
    MPF_REQUIRE(m.d_frame && m.d_frame_dyn && m.d_mask && m.d_mask_dyn && m.d_flow && m.d_flow_dyn && m.d_obj_mask && m.d_flow_mix && m.d_frame_mix && m.d_fill_mask,
                "mpf_merge_ex: null pointer");
    MPF_REQUIRE(m.obj_mask_stride >= 0 && m.obj_mask_stride <= 4, "mpf_merge_ex: obj_mask_stride must be 0..4");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_merge, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m, N);
    return mpf_launch_status("k_merge");
}

__global__ void __launch_bounds__(256)
k_to_u8_bgr(const float *__restrict__ img, int64_t N, uint8_t *__restrict__ out)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3 * n + c] = mpf_to_u8(img[(2 - c) * N + n]);
}

extern "C" int mpf_to_u8_bgr(const float *d_img, int H, int W, uint8_t *d_out, void *stream)
{
    MPF_REQUIRE(d_img && d_out && H >= 1 && W >= 1, "mpf_to_u8_bgr: bad argument");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_to_u8_bgr, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_img, N, d_out);
    return mpf_launch_status("k_to_u8_bgr");
}
