/* oracle_mpi.c - CPU restatement of utils/mpi (homography warp, volume rendering, flow) and of the merge step of
 * utils/utils.py.  TEST INFRASTRUCTURE ONLY - see oracle.h.
 *
 * "Numerics ledger": the reference's arithmetic is executed by PyTorch-CPU ATen kernels; what each of them does
 * to an fp32 value, as established by running the reference in the build container (tests/golden/make_golden.py;
 * tests/test_oracle_golden.py asserts every line below against the recorded tensors):
 *   L1  torch.matmul [S,3,3]x[S,3,N] (and 3x4 / 4x4 variants)  == k-ordered chain  a0*x, fma(a1,y,.), fma(a2,z,.)
 *   L2  tensor / tensor, tensor / python-scalar                   == IEEE fp32 division (no reciprocal-multiply)
 *   L3  F.grid_sample(bilinear, border, align_corners=False)      == x = (g+1)*(W/2) - 0.5, clamp to [0,W-1],
 *                                                                    weights nw=(1-fy)(1-fx) ..., value chain
 *                                                                    v_nw*nw, fma(v_ne,ne,.), fma(v_sw,sw,.), fma(v_se,se,.)
 *                                                                    with out-of-range neighbours read as 0
 *   L4  torch.norm(dim=2) over 3 components                       == sqrt(fma(z,z,fma(y,y,x*x)))
 *   L5  torch.cumprod (fp32, CPU)                                 == running product kept in double, emitted as float
 *   L6  torch.sum(dim=1) over S planes                            == ATen cascade sum: level-0 accumulates 16 addends,
 *                                                                    then is folded into level 1 (4 levels)
 *   L7  torch.exp                                                 == MKL VML, not reproducible: see oracle_math.c
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ---- small helpers ------------------------------------------------------------------------------------ */

/* L1: row . (x, y, 1)   (utils/mpi/homography_sampler.py:132-133, 208-209; mpi_rendering.py:234) */
static inline float row3_xy1(const float *r, float x, float y)
{
    float acc = r[0] * x;
    acc = fmaf(r[1], y, acc);
    acc = fmaf(r[2], 1.0f, acc);
    return acc;
}

/* L1: row of a 3x4 / 4x4 . (X, Y, Z, 1)   (utils/mpi/rendering_utils.py:18-19, geometry.py:67) */
static inline float row4_xyz1(const float *r, float X, float Y, float Z)
{
    float acc = r[0] * X;
    acc = fmaf(r[1], Y, acc);
    acc = fmaf(r[2], Z, acc);
    acc = fmaf(r[3], 1.0f, acc);
    return acc;
}

/* L4: torch.norm over 3 components (mpi_rendering.py:70, :106) */
static inline float norm3(float x, float y, float z)
{
    return sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
}

/* L6: at::native cascade sum (aten/src/ATen/native/cpu/SumKernel.cpp multi_row_sum), level_step = 16 for any
 * reduction length below 2^20.  Used by every torch.sum(dim=1) on the path (mpi_rendering.py:93-96, :132,
 * :143-152). */
typedef struct { float a[4]; int i; } csum_t;
static inline void csum_init(csum_t *c) { c->a[0] = c->a[1] = c->a[2] = c->a[3] = 0.0f; c->i = 0; }
static inline void csum_push(csum_t *c, float x)
{
    c->a[0] += x;
    c->i++;
    if ((c->i & 15) == 0) {
        for (int j = 1; j < 4; ++j) {
            c->a[j] += c->a[j - 1];
            c->a[j - 1] = 0.0f;
            if (c->i & (0xF << (4 * j))) break;
        }
    }
}
static inline float csum_final(const csum_t *c)
{
    float r = c->a[0];
    r += c->a[1]; r += c->a[2]; r += c->a[3];
    return r;
}

/* L2+L3: source coordinate (u,v) -> grid_sample taps.
 * homography_sampler.py:151-154 normalises ((u+.5)/(W*.5) - 1), ATen un-normalises ((g+1)*(W/2) - .5) and clamps
 * (padding_mode='border'); the round trip is kept because it perturbs the coordinate by a few ulp. */
typedef struct {
    float ix, iy;        /* clamped un-normalised coordinate */
    int x0, y0;          /* north-west texel */
    int e_in, s_in;      /* is x0+1 < W, is y0+1 < H */
    float nw, ne, sw, se;
} taps_t;

static inline void make_taps(float u, float v, int W, int H, taps_t *t)
{
    float gx = (u + 0.5f) / (float)(W * 0.5) - 1.0f;
    float gy = (v + 0.5f) / (float)(H * 0.5) - 1.0f;
    float ix = (gx + 1.0f) * ((float)W / 2.0f) - 0.5f;
    float iy = (gy + 1.0f) * ((float)H / 2.0f) - 0.5f;
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    float fx0 = floorf(ix), fy0 = floorf(iy);
    float w = ix - fx0, e = 1.0f - w;     /* distance to west / east  */
    float n = iy - fy0, s = 1.0f - n;     /* distance to north / south */
    t->ix = ix; t->iy = iy;
    t->x0 = (int)fx0; t->y0 = (int)fy0;
    t->e_in = (t->x0 + 1) < W;
    t->s_in = (t->y0 + 1) < H;
    t->nw = s * e; t->ne = s * w; t->sw = n * e; t->se = n * w;
}

static inline float bilerp(const taps_t *t, float v_nw, float v_ne, float v_sw, float v_se)
{
    float o = v_nw * t->nw;
    o = fmaf(v_ne, t->ne, o);
    o = fmaf(v_sw, t->sw, o);
    o = fmaf(v_se, t->se, o);
    return o;
}

static inline float sample_plane(const float *p, int W, const taps_t *t)
{
    const float *r0 = p + (int64_t)t->y0 * W + t->x0;
    float v_nw = r0[0];
    float v_ne = t->e_in ? r0[1] : 0.0f;
    float v_sw = t->s_in ? r0[W] : 0.0f;
    float v_se = (t->e_in && t->s_in) ? r0[W + 1] : 0.0f;
    return bilerp(t, v_nw, v_ne, v_sw, v_se);
}

/* ---- generic restatements ----------------------------------------------------------------------------- */

/* HomographySample.sample_inverse, utils/mpi/homography_sampler.py:197-218 (per-plane flow for every source pixel);
 * also the flowB2A by-product of .sample (:139-141) when given H_src_tgt. */
void orc_homography_flow(const float *hom, int S, int H, int W, float *flow)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < S; ++s)
        for (int y = 0; y < H; ++y) {
            const float *h = hom + 9 * s;
            float *o = flow + (((int64_t)s * H + y) * W) * 2;
            for (int x = 0; x < W; ++x) {
                float qx = row3_xy1(h, (float)x, (float)y);
                float qy = row3_xy1(h + 3, (float)x, (float)y);
                float qz = row3_xy1(h + 6, (float)x, (float)y);
                o[2 * x] = qx / qz - (float)x;
                o[2 * x + 1] = qy / qz - (float)y;
            }
        }
}

/* HomographySample.sample, utils/mpi/homography_sampler.py:124-158 (everything after H_src_tgt is known) */
void orc_homography_sample(const float *src, const float *hom, int S, int C, int H, int W,
                           float *tgt, uint8_t *valid, float *flowB2A)
{
    const int64_t N = (int64_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < S; ++s)
        for (int y = 0; y < H; ++y) {
            const float *h = hom + 9 * s;
            for (int x = 0; x < W; ++x) {
                float qx = row3_xy1(h, (float)x, (float)y);
                float qy = row3_xy1(h + 3, (float)x, (float)y);
                float qz = row3_xy1(h + 6, (float)x, (float)y);
                float u = qx / qz, v = qy / qz;
                int64_t pix = (int64_t)y * W + x;
                if (flowB2A) {
                    flowB2A[((int64_t)s * N + pix) * 2] = u - (float)x;
                    flowB2A[((int64_t)s * N + pix) * 2 + 1] = v - (float)y;
                }
                if (valid)
                    valid[(int64_t)s * N + pix] = (u < (float)W) && (u > -1.0f) && (v < (float)H) && (v > -1.0f);
                taps_t t;
                make_taps(u, v, W, H, &t);
                for (int c = 0; c < C; ++c)
                    tgt[((int64_t)s * C + c) * N + pix] = sample_plane(src + ((int64_t)s * C + c) * N, W, &t);
            }
        }
}

/* get_src_xyz_from_plane_disparity, utils/mpi/mpi_rendering.py:213-239 */
void orc_src_xyz(const float *k_inv, const float *depth, int S, int H, int W, float *xyz)
{
    const int64_t N = (int64_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int s = 0; s < S; ++s)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < 3; ++c)
                    xyz[((int64_t)s * 3 + c) * N + (int64_t)y * W + x] =
                        row3_xy1(k_inv + 3 * c, (float)x, (float)y) * depth[s];
}

/* transform_G_xyz, utils/mpi/rendering_utils.py:4-23 (via get_tgt_xyz_from_plane_disparity, mpi_rendering.py:242-256) */
void orc_transform_xyz(const float *G, const float *xyz, int S, int64_t N, float *out)
{
#pragma omp parallel for schedule(static)
    for (int s = 0; s < S; ++s) {
        const float *X = xyz + (int64_t)s * 3 * N, *Y = X + N, *Z = Y + N;
        float *o = out + (int64_t)s * 3 * N;
        for (int64_t n = 0; n < N; ++n)
            for (int c = 0; c < 3; ++c)
                o[c * N + n] = row4_xyz1(G + 4 * c, X[n], Y[n], Z[n]);
    }
}

/* plane_volume_rendering + weighted_sum_mpi, utils/mpi/mpi_rendering.py:62-99, :142-154
 * (plane_volume_rendering_flow :102-139 is the same chain with extra_in = per-plane flow) */
void orc_volume_render(const float *rgb, const float *sigma, const float *xyz, int S, int64_t N,
                       float *rgb_out, float *depth_out, float *tacc_out, float *weights_out,
                       const float *extra_in, int E, float *extra_out)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        double acc = 1.0;                                   /* L5 */
        csum_t cw, cd, cc[3], ce[8];
        csum_init(&cw); csum_init(&cd);
        for (int c = 0; c < 3; ++c) csum_init(&cc[c]);
        for (int e = 0; e < E && e < 8; ++e) csum_init(&ce[e]);
        for (int s = 0; s < S; ++s) {
            const float *p = xyz + (int64_t)s * 3 * N + n;
            float dist = 1e3f;                              /* :73-78 */
            if (s + 1 < S) {
                const float *q = p + 3 * N;
                dist = norm3(q[0] - p[0], q[N] - p[N], q[2 * N] - p[2 * N]);
            }
            float T = orc_expf(-sigma[(int64_t)s * N + n] * dist);   /* :79 */
            float alpha = 1.0f - T;                          /* :80 */
            float tacc = (float)acc;                         /* :84-88, exclusive */
            float w = tacc * alpha;                          /* :90 */
            acc *= (double)(T + 1e-6f);
            if (tacc_out) tacc_out[(int64_t)s * N + n] = tacc;
            if (weights_out) weights_out[(int64_t)s * N + n] = w;
            csum_push(&cw, w);
            if (rgb) for (int c = 0; c < 3; ++c) csum_push(&cc[c], w * rgb[((int64_t)s * 3 + c) * N + n]);
            csum_push(&cd, w * p[2 * N]);
            for (int e = 0; e < E && e < 8; ++e) csum_push(&ce[e], w * extra_in[((int64_t)s * E + e) * N + n]);
        }
        if (rgb_out) for (int c = 0; c < 3; ++c) rgb_out[c * N + n] = csum_final(&cc[c]);
        if (depth_out) depth_out[n] = csum_final(&cd) / (csum_final(&cw) + 1e-5f);   /* :152 */
        for (int e = 0; e < E && e < 8; ++e) extra_out[e * N + n] = csum_final(&ce[e]);
    }
}

/* alpha_composition (utils/mpi/mpi_rendering.py:42-59) and the blend weights of the use_alpha branch of render() (:36).
 * alpha [S,N], values [S,C,N] (may be NULL).  preserve_s = prod_{k<s} (1 - alpha_k)  (torch.cumprod: double accumulator,
 * L5), weights_s = alpha_s * preserve_s, out_c = cascade-sum_s values_sc * weights_s (L6);
 * cumprod_eps_s = prod_{k<=s} ((1 - alpha_k) + 1e-6)  (inclusive - "blend_weights"). */
void orc_alpha_composition(const float *alpha, const float *values, int S, int C, int64_t N, float *out, float *weights_out,
                           float *cumprod_eps_out)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        double keep = 1.0, keep_eps = 1.0;
        csum_t cc[8];
        for (int c = 0; c < C && c < 8; ++c) csum_init(&cc[c]);
        for (int s = 0; s < S; ++s) {
            const float a = alpha[(int64_t)s * N + n];
            const float w = a * (float)keep;
            keep *= (double)(1.0f - a);
            keep_eps *= (double)((1.0f - a) + 1e-6f);
            if (weights_out) weights_out[(int64_t)s * N + n] = w;
            if (cumprod_eps_out) cumprod_eps_out[(int64_t)s * N + n] = (float)keep_eps;
            if (values) for (int c = 0; c < C && c < 8; ++c) csum_push(&cc[c], values[((int64_t)s * C + c) * N + n] * w);
        }
        if (values && out) for (int c = 0; c < C && c < 8; ++c) out[c * N + n] = csum_final(&cc[c]);
    }
}

/* ---- streaming restatements of the fused stages ------------------------------------------------------- */

/* Stage A + C: utils/utils.py:190-204 (source-frame transmittance -> blend) fused with
 * HomographySample.sample_inverse + plane_volume_rendering_flow (mpi_rendering.py:102-139) for P poses. */
void orc_src_blend_flow(const float *mpi, const float *img, const float *k_inv, const float *depth,
                        const float *hom_ts, int P, int S, int H, int W, float flow_clip,
                        float *out_rgba, float *out_rgb_planar, float *out_tacc, float *flows)
{
    const int64_t N = (int64_t)H * W;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int64_t n = (int64_t)y * W + x;
            const float fx = (float)x, fy = (float)y;
            float ray[3];
            for (int c = 0; c < 3; ++c) ray[c] = row3_xy1(k_inv + 3 * c, fx, fy);
            float im[3] = { img[n], img[N + n], img[2 * N + n] };
            double acc = 1.0;
            csum_t cf[8][2];
            for (int p = 0; p < P && p < 8; ++p) { csum_init(&cf[p][0]); csum_init(&cf[p][1]); }
            float cur[3] = { ray[0] * depth[0], ray[1] * depth[0], ray[2] * depth[0] };
            for (int s = 0; s < S; ++s) {
                float dist = 1e3f;
                float nxt[3] = { 0, 0, 0 };
                if (s + 1 < S) {
                    for (int c = 0; c < 3; ++c) nxt[c] = ray[c] * depth[s + 1];
                    dist = norm3(nxt[0] - cur[0], nxt[1] - cur[1], nxt[2] - cur[2]);
                }
                const float *pl = mpi + (int64_t)s * 4 * N + n;
                float sg = pl[3 * N];
                float T = orc_expf(-sg * dist);
                float alpha = 1.0f - T;
                float tacc = (float)acc;
                float w = tacc * alpha;
                acc *= (double)(T + 1e-6f);
                if (out_tacc) out_tacc[(int64_t)s * N + n] = tacc;
                float one_m = 1.0f - tacc;
                for (int c = 0; c < 3; ++c) {
                    float a = tacc * im[c];
                    float b = one_m * pl[c * N];
                    float o = a + b;
                    if (out_rgba) out_rgba[((int64_t)s * N + n) * 4 + c] = o;
                    if (out_rgb_planar) out_rgb_planar[((int64_t)s * 3 + c) * N + n] = o;
                }
                if (out_rgba) out_rgba[((int64_t)s * N + n) * 4 + 3] = sg;
                for (int p = 0; p < P && p < 8; ++p) {
                    const float *h = hom_ts + ((int64_t)p * S + s) * 9;
                    float qx = row3_xy1(h, fx, fy), qy = row3_xy1(h + 3, fx, fy), qz = row3_xy1(h + 6, fx, fy);
                    csum_push(&cf[p][0], w * (qx / qz - fx));
                    csum_push(&cf[p][1], w * (qy / qz - fy));
                }
                for (int c = 0; c < 3; ++c) cur[c] = nxt[c];
            }
            for (int p = 0; p < P && p < 8; ++p)
                for (int k = 0; k < 2; ++k) {
                    float f = csum_final(&cf[p][k]);
                    if (flow_clip > 0.0f) f = fminf(fmaxf(f, -flow_clip), flow_clip);   /* utils/utils.py:348 */
                    flows[((int64_t)p * 2 + k) * N + n] = f;
                }
        }
}

typedef struct { float c[4]; float om; float xyz[3]; int valid; } warped_t;

static inline void xyz_tgt_at(const float *k_inv, const float *G, float d, float px, float py, float *out)
{
    float X = row3_xy1(k_inv, px, py) * d;
    float Y = row3_xy1(k_inv + 3, px, py) * d;
    float Z = row3_xy1(k_inv + 6, px, py) * d;
    for (int c = 0; c < 3; ++c) out[c] = row4_xyz1(G + 4 * c, X, Y, Z);
}

static inline void warp_one(const float *rgba, int interleaved, const float *obj_mask, const float *h,
                            const float *k_inv, const float *G, float d, int s, int H, int W, int x, int y,
                            int exact_xyz, warped_t *o)
{
    const int64_t N = (int64_t)H * W;
    float qx = row3_xy1(h, (float)x, (float)y);
    float qy = row3_xy1(h + 3, (float)x, (float)y);
    float qz = row3_xy1(h + 6, (float)x, (float)y);
    float u = qx / qz, v = qy / qz;
    o->valid = (u < (float)W) && (u > -1.0f) && (v < (float)H) && (v > -1.0f);
    taps_t t;
    make_taps(u, v, W, H, &t);
    if (interleaved) {
        const float *r0 = rgba + ((int64_t)s * N + (int64_t)t.y0 * W + t.x0) * 4;
        for (int c = 0; c < 4; ++c) {
            float v_nw = r0[c];
            float v_ne = t.e_in ? r0[4 + c] : 0.0f;
            float v_sw = t.s_in ? r0[(int64_t)W * 4 + c] : 0.0f;
            float v_se = (t.e_in && t.s_in) ? r0[(int64_t)W * 4 + 4 + c] : 0.0f;
            o->c[c] = bilerp(&t, v_nw, v_ne, v_sw, v_se);
        }
    } else {
        for (int c = 0; c < 4; ++c) o->c[c] = sample_plane(rgba + ((int64_t)s * 4 + c) * N, W, &t);
    }
    o->om = obj_mask ? sample_plane(obj_mask, W, &t) : 0.0f;
    if (exact_xyz) {
        float a[3], b[3] = { 0, 0, 0 }, c2[3] = { 0, 0, 0 }, e[3] = { 0, 0, 0 };
        xyz_tgt_at(k_inv, G, d, (float)t.x0, (float)t.y0, a);
        if (t.e_in) xyz_tgt_at(k_inv, G, d, (float)(t.x0 + 1), (float)t.y0, b);
        if (t.s_in) xyz_tgt_at(k_inv, G, d, (float)t.x0, (float)(t.y0 + 1), c2);
        if (t.e_in && t.s_in) xyz_tgt_at(k_inv, G, d, (float)(t.x0 + 1), (float)(t.y0 + 1), e);
        for (int c = 0; c < 3; ++c) o->xyz[c] = bilerp(&t, a[c], b[c], c2[c], e[c]);
    } else {
        xyz_tgt_at(k_inv, G, d, t.ix, t.iy, o->xyz);
    }
}

/* Stage B: HomographySample.sample (homography_sampler.py:124-158) + render_tgt_rgb_depth's composite
 * (mpi_rendering.py:336-347 -> plane_volume_rendering :62-99 -> weighted_sum_mpi :142-154), streamed per target
 * pixel without materialising any [S,...] intermediate. */
void orc_warp_composite(const float *rgba, int interleaved, const float *obj_mask,
                        const float *hom_st, const float *k_inv, const float *G, const float *depth,
                        int S, int H, int W, int exact_xyz,
                        float *rgb_out, float *depth_out, float *objmask_out, float *tgt_mask_out)
{
    const int64_t N = (int64_t)H * W;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int64_t n = (int64_t)y * W + x;
            warped_t cur, nxt;
            warp_one(rgba, interleaved, obj_mask, hom_st, k_inv, G, depth[0], 0, H, W, x, y, exact_xyz, &cur);
            double acc = 1.0;
            csum_t cw, cd, co, cc[3];
            csum_init(&cw); csum_init(&cd); csum_init(&co);
            for (int c = 0; c < 3; ++c) csum_init(&cc[c]);
            float nvalid = 0.0f;
            for (int s = 0; s < S; ++s) {
                float dist = 1e3f;
                if (s + 1 < S) {
                    warp_one(rgba, interleaved, obj_mask, hom_st + 9 * (s + 1), k_inv, G, depth[s + 1], s + 1,
                             H, W, x, y, exact_xyz, &nxt);
                    dist = norm3(nxt.xyz[0] - cur.xyz[0], nxt.xyz[1] - cur.xyz[1], nxt.xyz[2] - cur.xyz[2]);
                }
                float sg = (cur.xyz[2] >= 0.0f) ? cur.c[3] : 0.0f;       /* mpi_rendering.py:336-338 */
                float T = orc_expf(-sg * dist);
                float alpha = 1.0f - T;
                float tacc = (float)acc;
                float w = tacc * alpha;
                acc *= (double)(T + 1e-6f);
                csum_push(&cw, w);
                for (int c = 0; c < 3; ++c) csum_push(&cc[c], w * cur.c[c]);
                csum_push(&cd, w * cur.xyz[2]);
                csum_push(&co, w * cur.om);
                nvalid += cur.valid ? 1.0f : 0.0f;                        /* :347 */
                cur = nxt;
            }
            for (int c = 0; c < 3; ++c) rgb_out[c * N + n] = csum_final(&cc[c]);
            if (depth_out) depth_out[n] = csum_final(&cd) / (csum_final(&cw) + 1e-5f);
            if (objmask_out) objmask_out[n] = csum_final(&co);
            if (tgt_mask_out) tgt_mask_out[n] = nvalid;
        }
}

/* ---- Stage D ------------------------------------------------------------------------------------------ */

static inline uint8_t to_u8(float v)
{
    /* np.clip(np.round(x * 255), 0, 255).astype(np.uint8): np.round is round-half-to-even == rintf */
    float r = rintf(v * 255.0f);
    r = fminf(fmaxf(r, 0.0f), 255.0f);
    return (uint8_t)r;
}

/* utils/utils.py:174-177 / :237-242 : [3,H,W] float RGB -> [H,W,3] u8 BGR */
void orc_to_u8_bgr(const float *img, int H, int W, uint8_t *out)
{
    const int64_t N = (int64_t)H * W;
    for (int64_t n = 0; n < N; ++n)
        for (int c = 0; c < 3; ++c) out[n * 3 + c] = to_u8(img[(2 - c) * N + n]);
}

/* utils/utils.py:237-283 */
void orc_merge(const float *frame, const float *frame_dyn, const float *mask, const float *mask_dyn,
               const float *flow, const float *flow_dyn, const float *obj_mask, float th, int H, int W,
               float *flow_mix, uint8_t *frame_mix, uint8_t *fill_mask)
{
    const int64_t N = (int64_t)H * W;
    for (int64_t n = 0; n < N; ++n) {
        int obj = obj_mask[n] >= th;                    /* source-frame mask, :270-271, :277-278 */
        for (int k = 0; k < 2; ++k) {
            /* flow_np is zeroed where obj_mask < th, flow_dync where >= th; the select below never reads a
             * zeroed entry unless obj_mask is NaN (then flow_dync, un-zeroed, is kept: both tests false) */
            flow_mix[n * 2 + k] = obj ? flow[k * N + n] : flow_dyn[k * N + n];
        }
        int m = mask[n] >= th;                          /* target-frame masks, :273-276 */
        for (int c = 0; c < 3; ++c) {
            uint8_t a = (mask[n] < th) ? 255 : to_u8(frame[(2 - c) * N + n]);
            uint8_t b = (mask_dyn[n] < th) ? 255 : to_u8(frame_dyn[(2 - c) * N + n]);
            frame_mix[n * 3 + c] = m ? a : b;
        }
        float f = m ? 1.0f : mask_dyn[n];               /* :280-283 */
        fill_mask[n] = (f < th) ? 1 : 0;
    }
}


/* ---- input stage ------------------------------------------------------------------------------------
 * F.interpolate(x, size=(H,W), mode='bilinear', align_corners=True) as torch-CPU computes it for fp32 NCHW input with more
 * than one thread (gen_3dphoto_dynamic_v2.py:86-89, :104-105).  ATen has two kernels: out_H + out_W <= 128 takes the
 * "channels-last" one (combined weights, 4-term fma chain), everything larger the generic one (row interpolation, then
 * column).  Both established against torch 2.10 in this container (tests/golden/make_golden.py: input_stage). */
typedef struct { int i0, i1; float l0, l1; } OrcAxisTap;

static OrcAxisTap orc_axis_tap(int dst, int in, int out, float scale)
{
    OrcAxisTap t;
    if (in == out) { t.i0 = t.i1 = dst; t.l0 = 1.0f; t.l1 = 0.0f; return t; }
    const float src = scale * (float)dst;
    t.i0 = (int)floorf(src);
    if (t.i0 > in - 1) t.i0 = in - 1;
    t.l1 = src - (float)t.i0;
    if (t.l1 < 0.0f) t.l1 = 0.0f;
    if (t.l1 > 1.0f) t.l1 = 1.0f;
    t.l0 = 1.0f - t.l1;
    t.i1 = t.i0 + ((t.i0 < in - 1) ? 1 : 0);
    return t;
}

void orc_resize_bilinear_ac(const float *src, int C, int h, int w, int H, int W, float *out)
{
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    const int small = (H + W) <= 128;
    for (int c = 0; c < C; ++c) {
        const float *s = src + (int64_t)c * h * w;
        float *o = out + (int64_t)c * H * W;
#pragma omp parallel for
        for (int Y = 0; Y < H; ++Y) {
            const OrcAxisTap ty = orc_axis_tap(Y, h, H, sy);
            for (int X = 0; X < W; ++X) {
                const OrcAxisTap tx = orc_axis_tap(X, w, W, sx);
                const float v00 = s[(int64_t)ty.i0 * w + tx.i0], v01 = s[(int64_t)ty.i0 * w + tx.i1];
                const float v10 = s[(int64_t)ty.i1 * w + tx.i0], v11 = s[(int64_t)ty.i1 * w + tx.i1];
                float r;
                if (small) {
                    r = v01 * (ty.l0 * tx.l1);
                    r = fmaf(v00, ty.l0 * tx.l0, r);
                    r = fmaf(v10, ty.l1 * tx.l0, r);
                    r = fmaf(v11, ty.l1 * tx.l1, r);
                } else {
                    const float t0 = fmaf(v00, tx.l0, v01 * tx.l1);
                    const float t1 = fmaf(v10, tx.l0, v11 * tx.l1);
                    r = fmaf(t0, ty.l0, t1 * ty.l1);
                }
                o[(int64_t)Y * W + X] = r;
            }
        }
    }
}
