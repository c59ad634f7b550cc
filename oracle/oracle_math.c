/* oracle_math.c - scalar fp32 maths shared by the oracle's restatements.  TEST INFRASTRUCTURE ONLY. */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle.h"

static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* 2^q for an int q in the normal range, as SLEEF's pow2if */
static inline float pow2i(int q) { return bits_to_float((uint32_t)(q + 0x7f) << 23); }

/* transparency = torch.exp(-sigma * dist)   (utils/mpi/mpi_rendering.py:79 and :115).
 *
 * torch.exp on a CPU float tensor is NOT reproducible bit-for-bit: torch 2.10 (this image, MKL build) routes it
 * to Intel MKL VML vsExp in high-accuracy mode (measured here: identical results under
 * ATEN_CPU_CAPABILITY=avx512/avx2/default, 0.5-ulp class).  MKL is closed source, so the oracle offers two
 * stand-ins, selected with orc_set_exp_mode():
 *   mode 0 (default, "reference-like"): (float)exp((double)x) - correctly rounded in all but ~1e-8 of cases;
 *           differs from torch.exp by 1 ulp on ~2 % of inputs (measured, 2e6 samples).
 *   mode 1 ("kernel-like"): the fp32 algorithm the HIP kernels use (device function mpf_expf in
 *           mpiflow_amd/csrc/mpf_math.h): SLEEF expf_u10's published scheme - q = rint(x*log2e), two-step
 *           Cody-Waite reduction with fused multiply-adds, degree-6 Horner polynomial with fused multiply-adds,
 *           1 + (s*s*u + s), ldexp as two exact power-of-two multiplies.  <= 1 ulp; differs from torch.exp by
 *           1 ulp on ~8 % of inputs.  Because every step is an IEEE fp32 op the GPU reproduces it exactly, which
 *           lets tests compare the HIP kernels with the oracle bit-for-bit.
 * Either way the effect on weights/rgb/flow is ~1e-7, three orders below the 1e-4 parity tolerance. */
static int g_exp_mode = 0;
void orc_set_exp_mode(int mode) { g_exp_mode = mode; }
int  orc_get_exp_mode(void) { return g_exp_mode; }

static float expf_kernel_like(float d);

float orc_expf(float d)
{
    if (g_exp_mode == 0) return (float)exp((double)d);
    return expf_kernel_like(d);
}

static float expf_kernel_like(float d)
{
    const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float L2Uf = 0.693145751953125f;
    const float L2Lf = 1.428606765330187045e-06f;
    int q = (int)lrintf(d * R_LN2f);          /* default rounding mode: nearest-even, as cvtps2dq */
    float qf = (float)q;
    float s = fmaf(qf, -L2Uf, d);
    s = fmaf(qf, -L2Lf, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = 1.0f + fmaf(s * s, u, s);
    u = u * pow2i(q >> 1) * pow2i(q - (q >> 1));
    if (d < -104.0f) u = 0.0f;
    if (d > 100.0f) u = INFINITY;
    return u;
}

void orc_expf_array(const float *x, float *y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = orc_expf(x[i]);
}
