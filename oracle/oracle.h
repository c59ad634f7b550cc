/* oracle.h - CPU restatement ("oracle") of the MPI-Flow hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and only as
 * the checker.  The product (mpiflow_amd/, libmpiflow_hip.so) never links, imports or calls anything here.
 *
 * Every function restates, in plain C with every fp32 rounding written out (explicit fmaf, -ffp-contract=off),
 * the arithmetic the reference performs through PyTorch-CPU ATen kernels; the reference file:line each one
 * follows is cited at its definition.  Parity is PINNED: tests/test_oracle_golden.py checks this library
 * against tests/golden/*.npz, which tests/golden/make_golden.py produced by importing and running the reference
 * itself (torch 2.10.0 CPU, this image).
 *
 * Conventions: row-major host pointers, fp32 unless stated, B == 1 everywhere (as in the reference's entry
 * point), S planes ordered near -> far, N = H*W.
 */
#ifndef MPIFLOW_ORACLE_H
#define MPIFLOW_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- oracle_math.c ---------------------------------------------------------------------------------- */
void  orc_set_exp_mode(int mode);   /* 0: (float)exp((double)x) "reference-like"; 1: fp32 scheme of the HIP kernels */
int   orc_get_exp_mode(void);
float orc_expf(float x);
void  orc_expf_array(const float *x, float *y, int64_t n);

/* ---- oracle_mpi.c : generic (materialised-tensor) restatements of utils/mpi ---------------------------- */

/* per-plane 3x3 . (x,y,1) then perspective divide; flow = result - (x,y).  hom: [S,9].  out: [S,H,W,2] */
void orc_homography_flow(const float *hom, int S, int H, int W, float *flow_SHW2);

/* HomographySample.sample: warp C channels of every plane with its own H_src_tgt [S,9].
 * src [S,C,H,W] -> tgt [S,C,H,W], valid u8 [S,H,W], flowB2A [S,H,W,2] (may be NULL) */
void orc_homography_sample(const float *src, const float *hom_src_tgt, int S, int C, int H, int W,
                           float *tgt, uint8_t *valid, float *flowB2A);

/* get_src_xyz_from_plane_disparity: xyz[s,c,y,x] = (Kinv . (x,y,1))_c * depth[s] -> [S,3,H,W] */
void orc_src_xyz(const float *k_inv9, const float *depth_S, int S, int H, int W, float *xyz_S3HW);

/* transform_G_xyz (rows 0..2 of a 4x4 G) applied to [S,3,N] -> [S,3,N] */
void orc_transform_xyz(const float *G16, const float *xyz_S3N, int S, int64_t N, float *out_S3N);

/* plane_volume_rendering (+ optional flow / obj-mask sums).  Any of the optional pointers may be NULL.
 * rgb [S,3,N], sigma [S,1,N], xyz [S,3,N];  outputs: rgb_out [3,N], depth_out [N], tacc [S,N] (transparency_acc
 * = "blend_weights"), weights [S,N];  extra_in [S,E,N] summed with the weights into extra_out [E,N]. */
void orc_volume_render(const float *rgb, const float *sigma, const float *xyz, int S, int64_t N,
                       float *rgb_out, float *depth_out, float *tacc_out, float *weights_out,
                       const float *extra_in, int E, float *extra_out);

/* alpha_composition + the use_alpha blend weights (utils/mpi/mpi_rendering.py:42-59, :36); C <= 8 */
void orc_alpha_composition(const float *alpha, const float *values, int S, int C, int64_t N, float *out, float *weights_out,
                           float *cumprod_eps_out);

/* ---- oracle_mpi.c : streaming restatements of the fused stages (what the HIP kernels implement) -------- */

/* Stage A + C.  mpi [S,4,H,W] planar (rgb, sigma), img [3,H,W].  For every source pixel: transmittance chain on
 * analytic xyz_src, blend rgb, and P volume-rendered flows (hom_tgt_src [P,S,9]).
 * out_rgba: [S,H,W,4] interleaved blended rgb + sigma (may be NULL); out_rgb_planar [S,3,H,W] (may be NULL);
 * out_tacc [S,H,W] (may be NULL); flows [P,2,H,W] (clipped to +-flow_clip if flow_clip > 0). */
void orc_src_blend_flow(const float *mpi, const float *img, const float *k_inv9, const float *depth_S,
                        const float *hom_tgt_src, int P, int S, int H, int W, float flow_clip,
                        float *out_rgba, float *out_rgb_planar, float *out_tacc, float *flows);

/* Stage B.  rgba [S,H,W,4] interleaved if interleaved != 0, else planar [S,4,H,W].  obj_mask [H,W] or NULL.
 * exact_xyz != 0: warped xyz = bilinear sample of the per-texel xyz_tgt (bit-faithful to the reference);
 * exact_xyz == 0: xyz_tgt evaluated analytically at the clamped source coordinate.
 * Outputs: rgb [3,H,W], depth [H,W], objmask [H,W] (NULL if obj_mask NULL), tgt_mask [H,W]. */
void orc_warp_composite(const float *rgba, int interleaved, const float *obj_mask,
                        const float *hom_src_tgt, const float *k_inv9, const float *G16, const float *depth_S,
                        int S, int H, int W, int exact_xyz,
                        float *rgb_out, float *depth_out, float *objmask_out, float *tgt_mask_out);

/* Stage D merge (utils/utils.py:237-283).  Frames are [3,H,W] fp32 RGB, masks [H,W].
 * Outputs: flow_mix [H,W,2] f32, frame_mix [H,W,3] u8 BGR, fill_mask [H,W] u8. */
void orc_merge(const float *frame, const float *frame_dyn, const float *mask, const float *mask_dyn,
               const float *flow, const float *flow_dyn, const float *obj_mask, float thresh, int H, int W,
               float *flow_mix, uint8_t *frame_mix, uint8_t *fill_mask);

/* float image [3,H,W] in 0..1 -> u8 BGR [H,W,3] via clip(rint(x*255)) (utils/utils.py:174-177,240-242) */
void orc_to_u8_bgr(const float *img_3HW, int H, int W, uint8_t *out_HW3);

/* ---- oracle_fwarp.c : geometry.py + moving_obj.py + warping.c ------------------------------------------ */

/* depth -> pixel coordinates and depth in a second view (BackprojectDepth + Project3D), pixel units.
 * depth [H,W]; inv_K [9]; P = (K.T)[:3,:] as [12].  pix [H,W,2] in [-1,1]-normalised units as the reference
 * returns them; z [H,W]. */
void orc_backproject_project(const float *depth, const float *inv_k9, const float *P12, int H, int W,
                             float *pix_HW2, float *z_HW);

/* our own restatement of warping.c:6-33 (serial, order dependent) */
void orc_forward_warping(const uint8_t *src, const int64_t *idx, const int64_t *idy, const float *z,
                         uint8_t *warped, int h, int w);

/* moving_obj.py:108-153 after the two projections: select by instance mask, to pixel units, truncate+clamp.
 * p_static/p_obj [H,W,2] normalised, z_static/z_obj [H,W], inst [H,W].
 * outputs p1 [H,W,2] pixel units, z1 [H,W], safe_x/safe_y int64 [H,W], flow01 [H,W,2]. */
void orc_select_truncate(const float *p_static, const float *z_static, const float *p_obj, const float *z_obj,
                         const float *inst, int H, int W,
                         float *p1, float *z1, int64_t *safe_x, int64_t *safe_y, float *flow01);

/* masks from the warped array [H,W,5]: Hm valid, M collision, Md = dilate3x3(M), P = (Md == M), Hp = Hm*P */
void orc_warp_masks(const uint8_t *warped, int H, int W, uint8_t *Hm, uint8_t *M, uint8_t *Md, uint8_t *P,
                    uint8_t *Hp);

/* F.interpolate(size=(H,W), mode='bilinear', align_corners=True) on fp32 [C,h,w] -> [C,H,W] (gen_3dphoto_dynamic_v2.py:86-89,104-105) */
void orc_resize_bilinear_ac(const float *src, int C, int h, int w, int H, int W, float *out);

/* ---- oracle_inpaint.c : cv2.inpaint (NS / Telea) and cv2.dilate(3x3) restated - PARITY UNPINNED (third-party OpenCV) --- */
/* img u8 [rows,cols,C] (C = 1 or 3), mask u8 [rows,cols] (non-zero = fill), method 0 = INPAINT_NS, 1 = INPAINT_TELEA
 * (reference: utils/utils.py:284-286, moving_obj.py:162).  Returns 0, -1 on allocation failure. */
int orc_inpaint(const uint8_t *img, const uint8_t *mask, int rows, int cols, int C, double radius, int method, uint8_t *out);
/* 0 (default) = OpenCV's unqualified sqrt / fabs on float arguments taken as the float overloads, 1 = as the double functions
 * (see oracle_inpaint.c); an open question until the real cv2 has been run on tests/golden/inpaint_reading_exhibit.npz */
void orc_inpaint_set_reading(int reading);
/* cv2.dilate(img, ones(3,3)) on one u8 channel (moving_obj.py:144-145) */
void orc_dilate3x3(const uint8_t *img, int rows, int cols, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
