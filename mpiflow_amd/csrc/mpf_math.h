// mpf_math.h - device-side fp32 arithmetic shared by all MPI-Flow HIP kernels (gfx950).
//
// Every helper spells out each IEEE rounding with explicit fmaf / * / + so that, compiled with
// -ffp-contract=off and correctly rounded divide/sqrt, the GPU performs exactly the fp32 operation sequence the
// reference's PyTorch-CPU path performs (the "numerics ledger": DESIGN.md §3; established against the reference in
// tests/golden/).  The 1e-4 flow/RGB parity target is tighter than fp32 re-association noise amplified by the
// composite, so operation ORDER is part of the contract here, not an implementation detail.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MPF_DEV __device__ __forceinline__

// torch.matmul [.,3,3] x (x,y,1)  ==  a0*x, fma(a1,y,.), fma(a2,1,.)
// (reference: utils/mpi/homography_sampler.py:132-133, :208-209; utils/mpi/mpi_rendering.py:234; geometry.py:42)
MPF_DEV float mpf_row3_xy1(float a0, float a1, float a2, float x, float y)
{
    float acc = a0 * x;
    acc = fmaf(a1, y, acc);
    acc = fmaf(a2, 1.0f, acc);
    return acc;
}

// torch.matmul [.,3|4,4] x (X,Y,Z,1)   (reference: utils/mpi/rendering_utils.py:18-19; geometry.py:67)
MPF_DEV float mpf_row4_xyz1(float a0, float a1, float a2, float a3, float X, float Y, float Z)
{
    float acc = a0 * X;
    acc = fmaf(a1, Y, acc);
    acc = fmaf(a2, Z, acc);
    acc = fmaf(a3, 1.0f, acc);
    return acc;
}

// torch.norm(dim=2) over 3 components (reference: utils/mpi/mpi_rendering.py:70, :106)
MPF_DEV float mpf_norm3(float x, float y, float z)
{
    return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}

// exp for transparency = exp(-sigma*dist) (reference: utils/mpi/mpi_rendering.py:79, :115).
// The reference's torch.exp is MKL VML (closed, <1 ulp).  This is the published SLEEF expf_u10 scheme in plain
// IEEE fp32 ops (<= 1 ulp): round-to-nearest-even range reduction, two-step Cody-Waite with fma, degree-6 Horner
// with fma, exact two-step ldexp.  Every operation is exactly rounded, hence bit-reproducible on CPU and GPU:
// oracle/oracle_math.c mode 1 is the same sequence and tests compare kernels with it bit for bit.
MPF_DEV float mpf_expf(float d)
{
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f;
    const float L2Lf = 1.428606765330187045e-06f;
    float qf = rintf(d * R_LN2f);
    int q = (int)qf;
    float s = fmaf(qf, -L2Uf, d);
    s = fmaf(qf, -L2Lf, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    int q1 = q >> 1;
    u = u * __int_as_float((q1 + 0x7f) << 23) * __int_as_float((q - q1 + 0x7f) << 23);
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = __int_as_float(0x7f800000);
    return u;
}

// ---- normal-range fast paths ------------------------------------------------------------------------------------
// hipcc expands an IEEE fp32 division into v_div_scale x2, v_rcp, a 6-fma Newton/residual chain, v_div_fmas and
// v_div_fixup (11 VALU ops); the scale/fixup ops only act on denormal / huge-exponent-gap / zero / inf / NaN operands.
// Homography denominators are O(1) and the grid normalisation divides by W/2, so the hot kernels run the SAME fma chain
// without the range guards - bit-identical quotients for normal-range operands - and share the refined reciprocal
// between the quotients that have a common denominator.
MPF_DEV float mpf_rcp_nr(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    float e = fmaf(-d, r, 1.0f);
    return fmaf(e, r, r);
}
MPF_DEV float mpf_div_nr(float n, float d, float r /* = mpf_rcp_nr(d) */)
{
    float q = n * r;
    float e = fmaf(-d, q, n);
    q = fmaf(e, r, q);
    e = fmaf(-d, q, n);
    return fmaf(e, r, q);
}
// Division by a divisor whose CORRECTLY ROUNDED reciprocal y = RN(1/d) is at hand (a per-launch constant: half the image width /
// height): Markstein's sequence - q = RN(n*y), exact residual by fma, one correction - returns RN(n/d) for every n whose quotient
// is a normal number, so the second correction of mpf_div_nr (needed there because v_rcp + one Newton step is only ALMOST always
// the correctly rounded reciprocal) can go.  tools/div_const_exhaustive.hip: 0 mismatches against the double-precision quotient
// over every divisor k/2, k = 2..16384, and every numerator of 32 binades (70 binades for the image sizes in use).
MPF_DEV float mpf_div_by_const(float n, float d, float y /* = RN(1/d) */)
{
    float q = n * y;
    float e = fmaf(-d, q, n);
    return fmaf(e, y, q);
}
// correctly rounded sqrt for x in 2^-102 .. 2^127: v_rsq_f32, one coupled Newton step on (sqrt, 1/(2 sqrt)) and one residual
// correction - 1 transcendental, 2 multiplies, 5 fmas, all of them full-rate ops.  (Round 1 used v_sqrt_f32 followed by the +-1 ulp
// residual test hipcc emits for an IEEE sqrt: 2 integer adds, 2 compares and 2 selects, which issue at little more than half the
// rate of an fma on gfx950.)  tools/sqrt_exhaustive.hip compares both with the double-precision square root rounded to float on
// EVERY float: identical and correctly rounded on all 2^23 values of every binade from 2^-102 up; below that the residual
// underflows in both (the guarded library sqrt pre-scales there) - the kernels take square roots of squared plane distances, 1e-4 .. 1e7.
// x == 0 (two adjacent planes with EQUAL disparity - a user-supplied stack, or fp16-quantised plane disparities): v_rsq(0) is +inf
// and 0 * inf would turn the pixel into NaN where the reference gets dist = 0, T = exp(-0) = 1.  Clamping r to FLT_MAX (one
// v_min_f32; the identity for every x > 0, so the exhaustive check above is unaffected) makes every later product an exact 0:
// g = 0, e = 0.5, h finite, d = 0 -> returns +0, the correctly rounded root.
MPF_DEV float mpf_sqrt_nr(float x)
{
    const float r = fminf(__builtin_amdgcn_rsqf(x), 3.402823466e+38f);
    float g = x * r;
    float h = 0.5f * r;
    const float e = fmaf(-h, g, 0.5f);
    h = fmaf(h, e, h);
    g = fmaf(g, e, g);
    const float d = fmaf(-g, g, x);
    return fmaf(d, h, g);
}
MPF_DEV float mpf_norm3_nr(float x, float y, float z)
{
    return mpf_sqrt_nr(fmaf(z, z, fmaf(y, y, x * x)));
}
// same value as mpf_expf for every input: the argument clamp replaces the two range selects (exp(-105) already rounds
// to 0, exp(100) already overflows to +inf) and v_ldexp_f32 replaces the two exact power-of-two multiplies
MPF_DEV float mpf_expf_fast(float d)
{
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f;
    const float L2Lf = 1.428606765330187045e-06f;
    d = fminf(fmaxf(d, -105.0f), 100.0f);
    float qf = rintf(d * R_LN2f);
    float s = fmaf(qf, -L2Uf, d);
    s = fmaf(qf, -L2Lf, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    return ldexpf(u, (int)qf);
}

// ATen's cascade sum behind every torch.sum(dim=1) over the S planes (aten/src/ATen/native/cpu/SumKernel.cpp
// multi_row_sum; reference call sites utils/mpi/mpi_rendering.py:93-96, :132, :143-152): level 0 takes 16 addends,
// then folds into level 1, every 256 into level 2.  NL = 2 covers S < 256, NL = 3 covers S < 4096.
template <int NL>
struct MpfCsum {
    float a[NL];
    MPF_DEV void init()
    {
#pragma unroll
        for (int j = 0; j < NL; ++j) a[j] = 0.0f;
    }
    MPF_DEV void push(float x) { a[0] += x; }
    // call after the (i)-th push when (i & 15) == 0
    MPF_DEV void fold(int i)
    {
        a[1] += a[0];
        a[0] = 0.0f;
        if (NL > 2) {
            if ((i & 0xF0) == 0) {
                a[NL - 1] += a[1];
                a[1] = 0.0f;
            }
        }
    }
    MPF_DEV float final() const
    {
        float r = a[0];
#pragma unroll
        for (int j = 1; j < NL; ++j) r += a[j];
        return r;
    }
};

// grid_sample(bilinear, padding_mode='border', align_corners=False) taps for a source coordinate (u,v), including the
// reference's normalise (utils/mpi/homography_sampler.py:151-154) / ATen un-normalise round trip.
struct MpfTaps {
    float ix, iy;          // clamped un-normalised coordinate
    int x0, y0;            // north-west texel
    bool e_in, s_in;       // x0+1 < W, y0+1 < H
    float nw, ne, sw, se;
};

MPF_DEV MpfTaps mpf_make_taps(float u, float v, int W, int H)
{
    MpfTaps t;
    const float halfW = (float)W * 0.5f, halfH = (float)H * 0.5f;   // exact: python (W * 0.5) then float
    float gx = (u + 0.5f) / halfW - 1.0f;
    float gy = (v + 0.5f) / halfH - 1.0f;
    float ix = (gx + 1.0f) * halfW - 0.5f;
    float iy = (gy + 1.0f) * halfH - 0.5f;
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    float fx0 = floorf(ix), fy0 = floorf(iy);
    float w = ix - fx0, e = 1.0f - w;
    float n = iy - fy0, s = 1.0f - n;
    t.ix = ix; t.iy = iy;
    t.x0 = (int)fx0; t.y0 = (int)fy0;
    t.e_in = (t.x0 + 1) < W;
    t.s_in = (t.y0 + 1) < H;
    t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
    return t;
}

MPF_DEV float mpf_bilerp(const MpfTaps &t, float v_nw, float v_ne, float v_sw, float v_se)
{
    float o = v_nw * t.nw;
    o = fmaf(v_ne, t.ne, o);
    o = fmaf(v_sw, t.sw, o);
    o = fmaf(v_se, t.se, o);
    return o;
}

// np.clip(np.round(x*255), 0, 255).astype(uint8)  (reference: utils/utils.py:175, :240-242); np.round = half-to-even
MPF_DEV uint8_t mpf_to_u8(float v)
{
    float r = rintf(v * 255.0f);
    r = fminf(fmaxf(r, 0.0f), 255.0f);
    return (uint8_t)r;
}

// XCD-aware block remap (MI355X: 8 XCDs, block b is dispatched to XCD b % 8, each XCD has a private 4 MiB L2).
// Gives XCD k the k-th contiguous chunk of the logical tile order so that neighbouring tiles (which share source
// texel rows) hit the same L2.  Bijective for any grid size.  A pure speed choice: correctness never depends on it.
MPF_DEV unsigned mpf_xcd_remap(unsigned b, unsigned nwg)
{
    const unsigned NX = 8;
    unsigned xcd = b % NX, k = b / NX;
    unsigned q = nwg / NX, r = nwg % NX;
    unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
