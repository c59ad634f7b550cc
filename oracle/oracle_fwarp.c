/* oracle_fwarp.c - CPU restatement of the depth->flow projection (geometry.py), of the glue in moving_obj.py and of
 * the serial forward splat external/forward_warping/warping.c.  TEST INFRASTRUCTURE ONLY - see oracle.h.
 *
 * The real warping.c is also compiled, unmodified and from where it lies, into oracle/_ref/libwarping.so
 * (`make -C oracle ref`); tests/test_oracle_golden.py checks orc_forward_warping against it byte for byte. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* BackprojectDepth.forward (geometry.py:41-49) followed by Project3D.forward (geometry.py:63-76).
 *   cam = depth * (inv_K[:3,:3] . (x,y,1))          matmul == a0*x, fma(a1,y,.), fma(a2,1,.)   then fp32 multiply
 *   q   = P[3x4] . (cam, 1)                          matmul == k-ordered fma chain
 *   pix = q.xy / (q.z + 1e-7);  pix.x /= (w-1);  pix.y /= (h-1);  pix = (pix - 0.5) * 2
 * returns pix (normalised) and z = q.z (no eps). */
void orc_backproject_project(const float *depth, const float *inv_k, const float *P, int H, int W,
                             float *pix, float *z)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int64_t n = (int64_t)y * W + x;
            const float fx = (float)x, fy = (float)y;
            float cam[3];
            for (int c = 0; c < 3; ++c) {
                float a = inv_k[3 * c] * fx;
                a = fmaf(inv_k[3 * c + 1], fy, a);
                a = fmaf(inv_k[3 * c + 2], 1.0f, a);
                cam[c] = depth[n] * a;
            }
            float q[3];
            for (int c = 0; c < 3; ++c) {
                float a = P[4 * c] * cam[0];
                a = fmaf(P[4 * c + 1], cam[1], a);
                a = fmaf(P[4 * c + 2], cam[2], a);
                a = fmaf(P[4 * c + 3], 1.0f, a);
                q[c] = a;
            }
            float den = q[2] + 1e-7f;
            float px = q[0] / den, py = q[1] / den;
            px = px / (float)(W - 1);
            py = py / (float)(H - 1);
            pix[n * 2] = (px - 0.5f) * 2.0f;
            pix[n * 2 + 1] = (py - 0.5f) * 2.0f;
            z[n] = q[2];
        }
}

/* torch's float -> int64 on the reference's x86 host is cvttss2si: NaN, +-inf and |v| >= 2^63 give INT64_MIN (clamped to 0 next).
 * Written out so that the oracle does not lean on undefined behaviour (checked against torch-CPU: tests/test_oracle_golden.py). */
static int64_t trunc_like_x86(float v)
{
    return (v >= -9223372036854775808.0f && v < 9223372036854775808.0f) ? (int64_t)v : INT64_MIN;
}


/* moving_obj.py:108-124 and :153 */
void orc_select_truncate(const float *p_static, const float *z_static, const float *p_obj, const float *z_obj,
                         const float *inst, int H, int W,
                         float *p1, float *z1, int64_t *safe_x, int64_t *safe_y, float *flow01)
{
    const int64_t N = (int64_t)H * W;
    for (int64_t n = 0; n < N; ++n) {
        int sel = inst[n] > 0.0f;                                    /* :108-112 */
        float nx = sel ? p_obj[2 * n] : p_static[2 * n];
        float ny = sel ? p_obj[2 * n + 1] : p_static[2 * n + 1];
        z1[n] = sel ? z_obj[n] : z_static[n];
        float px = (nx + 1.0f) / 2.0f * (float)(W - 1);              /* :115-117 */
        float py = (ny + 1.0f) / 2.0f * (float)(H - 1);
        p1[2 * n] = px; p1[2 * n + 1] = py;
        int64_t tx = trunc_like_x86(px), ty = trunc_like_x86(py);   /* .long(): truncation toward zero, :121-122 */
        if (tx > W - 1) tx = W - 1;
        if (tx < 0) tx = 0;
        if (ty > H - 1) ty = H - 1;
        if (ty < 0) ty = 0;
        safe_x[n] = tx; safe_y[n] = ty;
        flow01[2 * n] = px - (float)(n % W);                         /* :153 */
        flow01[2 * n + 1] = py - (float)(n / W);
    }
}

/* external/forward_warping/warping.c:6-33 restated: sources visited in raster order; a source paints its target
 * if its z is smaller than the z of the PREVIOUS visitor of that target (1000 for the first), marks it valid,
 * records "was I the first visitor" in the collision byte, and always leaves its own z behind. */
void orc_forward_warping(const uint8_t *src, const int64_t *idx, const int64_t *idy, const float *z,
                         uint8_t *warped, int h, int w)
{
    const int64_t N = (int64_t)h * w;
    float *last_z = (float *)malloc(sizeof(float) * (size_t)N);
    for (int64_t n = 0; n < N; ++n) last_z[n] = 1000.0f;
    for (int64_t n = 0; n < N; ++n) {
        int64_t t = idy[n] * w + idx[n];
        if (z[n] < last_z[t]) {
            warped[t * 5] = src[n * 3];
            warped[t * 5 + 1] = src[n * 3 + 1];
            warped[t * 5 + 2] = src[n * 3 + 2];
        }
        warped[t * 5 + 3] = 1;
        warped[t * 5 + 4] = (last_z[t] == 1000.0f) ? 1 : 0;
        last_z[t] = z[n];
    }
    free(last_z);
}

/* moving_obj.py:133-150.  M = 1 - (collision == valid); M' = cv2.dilate(M, ones(3,3)) (zero beyond the border:
 * OpenCV's default morphology border never wins a max); P = (M' == M); H' = H * P */
void orc_warp_masks(const uint8_t *warped, int H, int W, uint8_t *Hm, uint8_t *M, uint8_t *Md, uint8_t *P,
                    uint8_t *Hp)
{
    const int64_t N = (int64_t)H * W;
    for (int64_t n = 0; n < N; ++n) {
        Hm[n] = warped[n * 5 + 3];
        M[n] = (uint8_t)(1 - (warped[n * 5 + 4] == warped[n * 5 + 3]));
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t m = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W && M[(int64_t)yy * W + xx] > m)
                        m = M[(int64_t)yy * W + xx];
                }
            int64_t n = (int64_t)y * W + x;
            Md[n] = m;
            P[n] = (uint8_t)(m == M[n]);
            Hp[n] = (uint8_t)(Hm[n] * P[n]);
        }
}
