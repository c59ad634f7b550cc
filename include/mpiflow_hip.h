/* mpiflow_hip.h - C ABI of libmpiflow_hip.so: the MI355X (gfx950) implementation of MPI-Flow's per-image hot path.
 *
 * The reference (Sharpiless/MPI-Flow) has no plugin/operator registry.  Its boundary for this path is
 *   (1) one real FFI symbol: `forward_warping` in external/forward_warping/warping.c:6, loaded with ctypes at
 *       moving_obj.py:12-13 and called with host pointers at moving_obj.py:127-129; and
 *   (2) plain Python call signatures (utils/utils.py, utils/mpi/ modules, geometry.py, moving_obj.py), whose arithmetic
 *       is carried out by PyTorch ATen kernels.
 * This library therefore exports (1) under its exact name and semantics, and for (2) one device-pointer entry point
 * per reference function on the path (the comment on each cites the reference lines it replaces).  The Python
 * package mpiflow_amd/ mirrors the reference's signatures and binds these symbols with ctypes; INTEGRATION.md shows
 * the same binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every mpf_* function returns 0 on success, otherwise a hipError_t value (or MPF_ERR_*); mpf_last_error()
 *     returns a description for the calling thread.  Nothing is thrown, nothing aborts.
 *   - pointers named d_* are DEVICE pointers (fp32 unless stated), row-major, contiguous; the caller owns them.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All calls are asynchronous on that
 *     stream and graph-capturable: no allocation, no synchronisation, no host<->device copies inside, except the
 *     two host-pointer conveniences at the bottom which say so.
 *   - small per-call matrices (K^-1, G, per-plane homographies, plane depths) arrive in ONE device buffer
 *     `d_params` of MPF_PARAMS_FLOATS(records) floats laid out as below; the host computes them with the
 *     reference's own batched torch-CPU expressions (bit-identical homographies are a parity requirement).
 *   - B == 1 (as in the reference's entry point); S planes ordered near -> far; N = H*W; S < 4096.
 */
#ifndef MPIFLOW_HIP_H
#define MPIFLOW_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPF_VERSION 601   /* round 3 (301): + mpf_warp_views_and_blend_next, mpf_warp_composite_split, mpf_src_flow, mpf_merge_depth_ordered;
                             round 4 (401): + mpf_moving_object_chain, mpf_warp_views_blend_next_merge_prev, mpf_stream_create_cu_subset / _destroy,
                             mpf_encoder_input, mpf_conv2d_f32, mpf_maxpool3x3s2_f32; MpfConvArgs + plane_major, loaders 4 / 5, epilogues 4 / 5 / 6;
                             round 5 (501): + the parity-grade producer engine mpf_pconv, mpf_pfmn_input, mpf_pencoder_input, mpf_pbilinear2x, mpf_pper_plane,
                             mpf_pplane_masks, mpf_pmaxpool3x3s2; MpfMergeArgs + obj_mask_stride, mpf_merge_ex, mpf_src_flow_hard;
                             (503): MpfConvArgs + bprime_table, pw (planes per workgroup of the few-block layers);
                             round 6 (601): + loader 6 (MPF_CONV_LD_NEAREST_PHASE), mpf_tune("fwarp_gate"), mpf_forward_warp_workspace + 256 bytes */

/* d_params layout (floats):
 *   [0..8]   K_src^-1 (3x3 row-major)            [9..20]  G_tgt_src rows 0..2 (3x4 row-major: R | t)
 *   [21..31] reserved (zero)
 *   then one 16-float record per plane (per plane and pose for mpf_src_blend_flow: record = s*P + p):
 *     [0..8] 3x3 homography (H_src_tgt for warps, H_tgt_src for flows)   [9] plane depth d_s = 1/disparity_s
 *     [10..15] reserved (zero)                                                                                  */
#define MPF_PARAMS_HEADER 32
#define MPF_PLANE_RECORD 16
#define MPF_PARAMS_FLOATS(records) (MPF_PARAMS_HEADER + MPF_PLANE_RECORD * (records))

#define MPF_ERR_BAD_ARGUMENT 10001
#define MPF_ERR_UNSUPPORTED  10002

int mpf_version(void);
const char *mpf_last_error(void);
/* fills CU count, HBM bytes, gfx arch name (e.g. "gfx950"); any pointer may be NULL */
int mpf_device_info(int device, int *cu_count, size_t *hbm_bytes, char *arch, size_t arch_len);
/* A stream restricted to every `stride`-th compute unit (starting at `offset`): hipExtStreamCreateWithCUMask.  For latency-sized side work
 * underneath a chip-filling launch on another stream (pipeline.OverlappedPairRenderer.attach_chain(cu_stride=...)).  The caller owns the
 * stream (mpf_stream_destroy). */
int mpf_stream_create_cu_subset(int stride, int offset, void **out_stream);
int mpf_stream_destroy(void *stream);

/* Scheduling knobs of the product library (libmpiflow_hip.so) - none of them can change a result, every setting gives the same bytes (asserted by the tests):
 *   "sbf_px" = pixels per thread of mpf_src_blend_flow (0 auto, 1, 2);  "conv_pf" = 0 | 1 (default): the plane-walking conv kernels (MpfConvArgs.pw > 1) copy the
 *   next step's weight fragments / raw tile into a second LDS buffer during the current MFMA phase;  "chain_grid" / "chain_prio" = grid cap / s_setprio of the
 *   forward-warp kernels;  "fwarp_gate" = bucket visits above which caller-supplied targets leave the gather path for the radix path (-1 = default: one visit per
 *   source; 0 = always radix).
 * Keys that select RETIRED KERNEL VARIANTS ("stage_b", "planar_lds", "ovl_depth", "ovl_xcd_a", "view_shift", "fwarp_path": bit-identical witnesses of the shipped
 * kernels) or TIMING ABLATIONS ("ovl_ablate", "stage_b" 101..106: INVALID results) exist only in the witness build, libmpiflow_hip_witness.so (-DMPF_WITNESS; the
 * tests and tools load it through mpiflow_amd._lib.witness()); the product library returns MPF_ERR_BAD_ARGUMENT for them and does not contain those kernels.
 * The knobs are PROCESS-GLOBAL plain ints, not per-stream and NOT thread-safe: set them from one thread while no other thread is launching work through this
 * library.  Unknown keys return MPF_ERR_BAD_ARGUMENT. */
int mpf_tune(const char *key, int value);
int mpf_is_witness_build(void);   /* 1 for libmpiflow_hip_witness.so, 0 for the product library */

/* ================= fused hot path =============================================================================== */

/* Stage A + C.  Replaces utils/utils.py:190-204 (get_src_xyz_from_plane_disparity + render() in the source frame +
 * the blend of the source image into every plane) fused with HomographySample.sample_inverse
 * (utils/mpi/homography_sampler.py:160-220) + plane_volume_rendering_flow (utils/mpi/mpi_rendering.py:102-139) for
 * P = 0, 1 or 2 poses sharing the same source-frame weights.
 *   d_mpi [S,4,H,W] planar (rgb, sigma);  d_img [3,H,W];  d_params with S*P records (S records if P == 0; only the
 *   depth is read then).  Outputs, each optional (NULL to skip):
 *   d_out_rgba [S,H,W,4] interleaved blended rgb + sigma (the layout mpf_warp_composite streams fastest);
 *   d_out_rgb_planar [S,3,H,W];  d_out_tacc [S,H,W] (= "blend_weights");  d_flows [P,2,H,W], clipped to
 *   +-flow_clip when flow_clip > 0 (utils/utils.py:348).
 *   Per-pixel by-products fused into the same pass (each NULL to skip): d_src_u8_bgr [H,W,3] = the source frame as uint8
 *   BGR (utils/utils.py:174-177);  d_quads / d_quads_complement = mpf_build_mask_quads(d_obj_mask, 0 / 1).
 *   d_cum_mask [S,H,W] (NULL = d_mpi is already activated): d_mpi then holds the RAW last-layer output of the AdaMPI
 *   decoder and the network's activation epilogue (model/CPN/decoder.py:166-173: rgb = sigmoid(x), sigma = relu(x * cum_mask)
 *   + 1e-4) is applied in registers while the stack is streamed. */
int mpf_src_blend_flow(const float *d_mpi, const float *d_img, const float *d_params, int P, int S, int H, int W,
                       float flow_clip, float *d_out_rgba, float *d_out_rgb_planar, float *d_out_tacc,
                       float *d_flows, uint8_t *d_src_u8_bgr, const float *d_obj_mask, float *d_quads,
                       float *d_quads_complement, const float *d_cum_mask, void *stream);

/* obj_mask [H,W] -> per-texel quads (m[y,x], m[y,x+1], m[y+1,x], m[y+1,x+1]) as float4 [H,W,4], out-of-range
 * neighbours 0, of (complement ? 1 - m : m): the four bilinear taps of the mask channel in one 16-byte load.
 * (utils/utils.py:328 repeats the same [H,W] mask over all S planes; :225 passes 1 - obj_mask.) */
int mpf_build_mask_quads(const float *d_obj_mask, int complement, int H, int W, float *d_quads, void *stream);

/* Stage B - the north-star kernel.  Replaces HomographySample.sample (utils/mpi/homography_sampler.py:80-158) on the
 * 8-channel stack + render_tgt_rgb_depth's composite (utils/mpi/mpi_rendering.py:336-347 -> plane_volume_rendering
 * :62-99 -> weighted_sum_mpi :142-154), streamed per target pixel; xyz_tgt is evaluated analytically at the clamped
 * source coordinate instead of being warped as 3 extra channels.
 *   d_rgba: planar [S,4,H,W] if interleaved == 0; [S,H,W,4] if interleaved == 1; interleaved == 2 is [S,H,W,4] with the
 *   promise that at least (W+1)*16 bytes of readable, FINITE padding follow the last plane (lets the east/south taps use
 *   fixed offsets; an out-of-image tap has weight exactly 0).  d_mask_quads from mpf_build_mask_quads or NULL;
 *   d_params with S records holding H_src_tgt.  Outputs: d_rgb [3,H,W], d_depth [H,W] (NULL ok),
 *   d_objmask [H,W] (NULL iff d_mask_quads NULL), d_tgt_mask [H,W] = number of planes whose source coordinate is in
 *   range (NULL ok), d_rgb_u8_bgr [H,W,3] = the rendered frame as uint8 BGR, clip(rint(x*255)) (utils/utils.py:240-242;
 *   NULL ok).  Passing NULL for both d_depth and d_tgt_mask selects a leaner kernel body. */
int mpf_warp_composite(const float *d_rgba, int interleaved, const float *d_mask_quads, const float *d_params,
                       int S, int H, int W, float *d_rgb, float *d_depth, float *d_objmask, float *d_tgt_mask,
                       uint8_t *d_rgb_u8_bgr, void *stream);

/* Stage B on the stack as render_novel_view_dynamic receives it (utils/utils.py:291-349: mpi_all_rgb_src [1,S,3,H,W] and
 * mpi_all_sigma_src [1,S,1,H,W], two separate channel-planar tensors): the body of mpf_warp_composite reading the three colour planes
 * and the sigma plane where they lie - a tap pair of one channel row is one 8-byte load - so the caller assembles nothing (the
 * reference concatenates 8 channels per plane, utils/mpi/mpi_rendering.py:288-301).  Outputs as mpf_warp_composite; bit-identical to it.
 * mpf_warp_composite(interleaved = 0) runs the same kernel on one [S,4,H,W] tensor. */
int mpf_warp_composite_split(const float *d_rgb_S3HW, const float *d_sigma_SHW, const float *d_mask_quads, const float *d_params,
                             int S, int H, int W, float *d_rgb, float *d_depth, float *d_objmask, float *d_tgt_mask,
                             uint8_t *d_rgb_u8_bgr, void *stream);

/* Stage C alone on a bare sigma tensor [S,H,W]: the volume-rendered flows of P = 1 or 2 poses (HomographySample.sample_inverse,
 * utils/mpi/homography_sampler.py:160-220, + plane_volume_rendering_flow, utils/mpi/mpi_rendering.py:102-139) - what
 * render_novel_view_dynamic needs besides the warp (utils/utils.py:340-348).  d_params as for mpf_src_blend_flow (S*P records);
 * d_flows [P,2,H,W], clipped to +-flow_clip when flow_clip > 0.  Same arithmetic as mpf_src_blend_flow's flows: bit-identical. */
int mpf_src_flow(const float *d_sigma_SHW, const float *d_params, int P, int S, int H, int W, float flow_clip, float *d_flows, void *stream);
/* hard_flow = True (utils/mpi/mpi_rendering.py:126-130): per pixel the flow of the plane with the largest rendering weight (the first one on ties, as
 * torch.argmax) instead of the weighted sum - one pass over the sigma planes, nothing per-plane materialised.  d_sigma: plane s at d_sigma + s *
 * plane_stride floats (H*W for a bare [S,H,W] tensor; 4*H*W with d_sigma = stack + 3*H*W for the [S,4,H,W] stack).  Weights with mpf_src_blend_flow's
 * arithmetic; equals mpf_homography_flow + mpf_volume_render(hard) bit for bit. */
int mpf_src_flow_hard(const float *d_sigma, int64_t plane_stride, const float *d_params, int P, int S, int H, int W, float flow_clip, float *d_flows,
                      void *stream);

/* Stage B for SEVERAL views of one stack in one launch.  The reference renders two poses of every stack
 * (utils/utils.py:210-222 with obj_mask / cam_ext and :224-236 with 1 - obj_mask / cam_ext_dynamic) and `repeat` such
 * pairs per image (gen_3dphoto_dynamic_v2.py:99-118); launched together, the views' workgroups walk the planes side by
 * side and the stack is fetched from HBM once per launch instead of once per view.  Results are bit-identical to n_views
 * calls of mpf_warp_composite.  `views` is a HOST array of n_views (<= MPF_MAX_VIEWS) descriptors holding DEVICE
 * pointers with the meaning of the same-named mpf_warp_composite arguments; it is copied into the kernel arguments, so
 * it may be reused or freed as soon as the call returns.  All views take a mask, or none does.  interleaved: 1 or 2. */
#define MPF_MAX_VIEWS 16
typedef struct MpfWarpView {
    const float *d_params;
    const float *d_mask_quads;
    float *d_rgb, *d_depth, *d_objmask, *d_tgt_mask;
    uint8_t *d_rgb_u8_bgr;
} MpfWarpView;
int mpf_warp_composite_views(const float *d_rgba, int interleaved, const MpfWarpView *views, int n_views, int S, int H,
                             int W, void *stream);

/* Stage B of one image AND Stage A+C of the NEXT image in one launch - the throughput form of the reference's unit of work
 * (utils/utils.py:190-236: one source-frame pass, :190-204 + render :7-39, then two target-frame passes, :210-236).  Run back to back
 * the two stages are an HBM-bound kernel followed by a VALU-issue-bound one; here a heterogeneous grid interleaves their workgroups
 * so that every CU holds both kinds at once.  Exactly the arithmetic of
 *     mpf_warp_composite_views(d_rgba, 2, views, n_views, S, H, W)                                              and
 *     mpf_src_blend_flow(d_mpi_next, d_img_next, d_params_next, P, S, H, W, flow_clip, d_out_rgba_next, NULL, NULL, d_flows_next,
 *                        d_src_u8_bgr_next, d_obj_mask_next, d_quads_next, d_quads_complement_next, d_cum_mask_next)
 * - results are bit-identical to those two calls; arguments have the meaning of their same-named counterparts there.
 * d_rgba (read) is a tail-padded interleaved stack (interleaved == 2); d_out_rgba_next (written) must be a DIFFERENT buffer, and so
 * must every *_next output be from anything the views read or write: the two halves of the launch are unordered. */
int mpf_warp_views_and_blend_next(const float *d_rgba, const MpfWarpView *views, int n_views,
                                  const float *d_mpi_next, const float *d_img_next, const float *d_params_next, int P,
                                  float flow_clip, float *d_out_rgba_next, float *d_flows_next, uint8_t *d_src_u8_bgr_next,
                                  const float *d_obj_mask_next, float *d_quads_next, float *d_quads_complement_next,
                                  const float *d_cum_mask_next, int S, int H, int W, void *stream);

/* The same launch with Stage D (mpf_merge) of an EARLIER pair folded in: the Stage A+C role runs it as a per-pixel prologue, so a stream
 * of pairs is ONE launch per pair with nothing between two launches (pipeline.OverlappedPairRenderer(merge_in_launch=True): pair i's
 * Stage A+C in launch i, its Stage B in launch i+1, its merge in launch i+2).  merge_prev = mpf_merge's arguments (NULL: plain
 * mpf_warp_views_and_blend_next).  Its d_flow / d_flow_dyn MAY be (parts of) d_flows_next - the thread that merges a pixel is the one that
 * later writes that pixel's new flows; its frames / masks must not be the views this launch renders. */
typedef struct MpfMergeArgs {
    const float *d_frame, *d_frame_dyn;        /* [3,H,W] the two rendered views */
    const float *d_mask, *d_mask_dyn;          /* [H,W] their rendered object masks */
    const float *d_flow, *d_flow_dyn;          /* [2,H,W] the two volume-rendered flows */
    const float *d_obj_mask;                   /* [H,W] source-frame object mask (see obj_mask_stride) */
    float thresh;
    float *d_flow_mix;                         /* [H,W,2] */
    uint8_t *d_frame_mix, *d_fill_mask;        /* [H,W,3] BGR, [H,W] */
    int obj_mask_stride;                       /* floats between consecutive pixels of d_obj_mask: 0 | 1 = a plain [H,W] map; 4 = the first component of a
                                                  mask-quad buffer [H,W,4] (what Stage A+C wrote for that pair: quads[n].x == obj_mask[n]) - lets a pipelined
                                                  caller merge a pair from buffers it owns, whatever happened to the caller's mask tensor since */
} MpfMergeArgs;
int mpf_warp_views_blend_next_merge_prev(const float *d_rgba, const MpfWarpView *views, int n_views,
                                         const float *d_mpi_next, const float *d_img_next, const float *d_params_next, int P,
                                         float flow_clip, float *d_out_rgba_next, float *d_flows_next, uint8_t *d_src_u8_bgr_next,
                                         const float *d_obj_mask_next, float *d_quads_next, float *d_quads_complement_next,
                                         const float *d_cum_mask_next, int S, int H, int W, const MpfMergeArgs *merge_prev, void *stream);

/* Stage D.  Replaces utils/utils.py:237-283 (uint8 BGR conversion, threshold, layer select, fill mask).
 * frames [3,H,W] RGB float, masks [H,W], flows [2,H,W], obj_mask [H,W] ->
 * d_flow_mix [H,W,2] f32, d_frame_mix [H,W,3] u8 BGR, d_fill_mask [H,W] u8 (1 = hole to inpaint). */
int mpf_merge(const float *d_frame, const float *d_frame_dyn, const float *d_mask, const float *d_mask_dyn,
              const float *d_flow, const float *d_flow_dyn, const float *d_obj_mask, float thresh, int H, int W,
              float *d_flow_mix, uint8_t *d_frame_mix, uint8_t *d_fill_mask, void *stream);

/* mpf_merge with its arguments as the struct (obj_mask_stride honoured) */
int mpf_merge_ex(const MpfMergeArgs *args, int H, int W, void *stream);

/* The depth-ordered variant of Stage D's frame ("utils/utils copy.py":278-303, the reference's older per-image module): frame_mix as
 * mpf_merge computes it, except that where both layers cover the pixel (both masks non-zero) and depth > depth_dyn the dynamic layer's
 * pixel is taken.  depths [H,W] are the two views' composited depths (mpf_warp_composite's d_depth).
 * -> d_frame_mix_depth [H,W,3] u8 BGR; d_depth_mask [H,W] u8 (optional, may be NULL): 1 where the dynamic layer was picked. */
int mpf_merge_depth_ordered(const float *d_frame, const float *d_frame_dyn, const float *d_mask, const float *d_mask_dyn,
                            const float *d_depth, const float *d_depth_dyn, float thresh, int H, int W,
                            uint8_t *d_frame_mix_depth, uint8_t *d_depth_mask, void *stream);

/* Built-in hole fill used when OpenCV is absent (NOT cv2.inpaint's Navier-Stokes / Telea, utils/utils.py:284-286,
 * moving_obj.py:162; row A13 is parity-unpinned): onion peel - pass k gives every hole pixel that touches
 * a pixel known after pass k-1 the rounded mean of those 8-neighbours.  In place on d_img u8 [H,W,3]; d_hole u8 [H,W]
 * (1 = hole) is cleared where filled (holes without any known pixel in their connected region stay 1).  One launch
 * sequence on the stream, no host round trip.  d_workspace: mpf_fill_holes_workspace(H, W) bytes. */
size_t mpf_fill_holes_workspace(int H, int W);
int mpf_fill_holes(uint8_t *d_img, uint8_t *d_hole, int H, int W, void *d_workspace, size_t workspace_bytes, void *stream);

/* The reference's own hole filling, on the HOST (host pointers, synchronous, re-entrant, no GPU involved): OpenCV's
 * cv2.inpaint(img, mask, radius, flags) for flags = cv2.INPAINT_NS (utils/utils.py:284-286: frame_mix, fill_mask, radius 3)
 * and cv2.INPAINT_TELEA (moving_obj.py:162: im1_raw, 1 - H, radius 3) - the fast-marching front of modules/photo/src/inpaint.cpp
 * with its order of operations (third-party arithmetic: parity with cv2 itself is unpinned until a test has run next to a real
 * cv2).  img u8 [H,W,C] (C = 1 or 3), mask u8 [H,W] (non-zero = fill), out u8 [H,W,C] (may not alias img).
 * Filling is sequential by nature (every pixel reads pixels filled before it), so callers run one frame per host thread. */
#define MPF_INPAINT_NS 0
#define MPF_INPAINT_TELEA 1
int mpf_inpaint_host(const uint8_t *img, const uint8_t *mask, int H, int W, int C, double radius, int method, uint8_t *out);

/* End-of-batch statistics of one pair, without a host round trip: d_out holds
 * MPF_PAIR_STATS_SLICES rows of 4 doubles, one per fixed contiguous slice of the frame:
 * { sum |flow|, hole pixels, max |flow|, max(-flow) } of d_flow_mix [H,W,2] f32 / d_fill_mask [H,W] u8 (empty slices: 0, 0,
 * -inf, -inf).  Sum the first two columns and take the maximum of the last two.  Deterministic summation order. */
#define MPF_PAIR_STATS_SLICES 64
int mpf_pair_stats(const float *d_flow_mix, const uint8_t *d_fill_mask, int H, int W, double *d_out, void *stream);

/* Measurement aid (bench.py `hbm_reference`; SURVEY.md 8(d) asks for an on-box streaming figure beside the 8 TB/s specification):
 * mode 0 reads `bytes` from d_src with 16-byte loads (d_dst: one float, never written for ordinary data), mode 1 copies d_src to
 * d_dst.  16-byte aligned device buffers, bytes a multiple of 16.  No counterpart in the reference. */
int mpf_stream_probe(const void *d_src, void *d_dst, size_t bytes, int mode, void *stream);

/* Frame -> PNG scanlines on the device (what cv2.imwrite does first, utils/utils.py:240-242 / gen_3dphoto_dynamic_v2.py:121-122):
 * d_bgr u8 [H,W,3] -> d_scanlines u8 [H, 1 + 3W]: filter byte 2 ("Up") followed by the RGB row minus the previous row
 * (mod 256).  The host only deflates these bytes and wraps them in chunks (mpiflow_amd/io_formats.py). */
int mpf_png_filter_up(const uint8_t *d_bgr, int H, int W, uint8_t *d_scanlines, void *stream);

/* [3,H,W] float RGB -> [H,W,3] u8 BGR, clip(rint(x*255))  (utils/utils.py:174-177) */
int mpf_to_u8_bgr(const float *d_img, int H, int W, uint8_t *d_out, void *stream);

/* Input stage.  Replaces, for one image: image_to_tensor / disparity_to_tensor after the file decode (utils/utils.py:35-52:
 * u8 / 255 in fp32 for the image, u8 / 255 in fp64 cast to fp32 for the disparity), the instance mask (ids == obj_index) as
 * float (gen_3dphoto_dynamic_v2.py:101-103) and the three F.interpolate(size=(H,W), mode='bilinear', align_corners=True) calls
 * (:86-89, :104-105), bit for bit as ATen's CPU kernels compute them.  d_rgb_u8 [h,w,3] -> d_image [3,H,W];
 * d_disp_u8 [h,w] -> d_disp [H,W]; d_ids_u8 [h,w] -> d_mask [H,W]; each pair optional (both NULL to skip). */
int mpf_prepare_inputs(const uint8_t *d_rgb_u8, const uint8_t *d_disp_u8, const uint8_t *d_ids_u8, int obj_index, int h, int w,
                       int H, int W, float *d_image, float *d_disp, float *d_mask, void *stream);

/* ================= generic (materialised-tensor) ops behind the utils/mpi function signatures ==================== */

/* get_src_xyz_from_plane_disparity (utils/mpi/mpi_rendering.py:213-239): params header K^-1 + S records (depth) */
int mpf_src_xyz(const float *d_params, int S, int H, int W, float *d_xyz_S3HW, void *stream);
/* transform_G_xyz (utils/mpi/rendering_utils.py:4-23): params header G; xyz [S,3,N] -> [S,3,N] */
int mpf_transform_xyz(const float *d_params, const float *d_xyz, int S, int64_t N, float *d_out, void *stream);
/* HomographySample.sample after H_src_tgt is known (utils/mpi/homography_sampler.py:124-158):
 * src [S,C,H,W] -> tgt [S,C,H,W], valid u8 [S,H,W] (NULL ok), flowB2A [S,H,W,2] (NULL ok) */
int mpf_homography_sample(const float *d_src, const float *d_params, int S, int C, int H, int W, float *d_tgt,
                          uint8_t *d_valid, float *d_flowB2A, void *stream);
/* HomographySample.sample_inverse after H_tgt_src is known (utils/mpi/homography_sampler.py:197-218): [S,H,W,2] */
int mpf_homography_flow(const float *d_params, int S, int H, int W, float *d_flow, void *stream);
/* plane_volume_rendering / plane_volume_rendering_flow / weighted_sum_mpi (utils/mpi/mpi_rendering.py:62-154) on
 * materialised rgb [S,3,N] (NULL ok), sigma [S,N], xyz [S,3,N]; extra_in [S,E,N] (E <= 4) is summed with the same
 * weights into extra_out [E,N] (flow and/or obj-mask).  hard != 0: extra is taken from the arg-max-weight plane only
 * (hard_flow, :126-130). */
int mpf_volume_render(const float *d_rgb, const float *d_sigma, const float *d_xyz, int S, int64_t N,
                      float *d_rgb_out, float *d_depth_out, float *d_tacc_out, float *d_weights_out,
                      const float *d_extra_in, int E, float *d_extra_out, int hard, void *stream);

/* weighted_sum_mpi's sums with caller-supplied weights (utils/mpi/mpi_rendering.py:143-152):
 * out[c,n] = cascade-sum_s weights[s,n] * values[s,c,n]   (values NULL: plain sum of the weights, C must be 1) */
int mpf_weighted_sum(const float *d_weights, const float *d_values, int S, int C, int64_t N, float *d_out, void *stream);

/* alpha_composition (utils/mpi/mpi_rendering.py:42-59) and the blend weights of render(use_alpha=True) (:36):
 * d_alpha [S,N]; d_values [S,C,N] with d_out [C,N] = cascade-sum_s values * weights (both NULL to skip);
 * d_weights [S,N] = alpha_s * prod_{k<s}(1 - alpha_k) (optional); d_cumprod_eps [S,N] = prod_{k<=s}(1 - alpha_k + 1e-6) (optional) */
int mpf_alpha_composite(const float *d_alpha, const float *d_values, int S, int C, int64_t N, float *d_out, float *d_weights,
                        float *d_cumprod_eps, void *stream);

/* ================= depth -> flow projection and forward warp (geometry.py, moving_obj.py, warping.c) ============= */

/* moving_obj.py:29-30: depth = 1 / (disp + 0.005), values above 100 clamped to 100 */
int mpf_disp_to_depth(const float *d_disp, int64_t N, float *d_depth, void *stream);
/* BackprojectDepth + Project3D (geometry.py:41-49, :63-76): depth [H,W]; inv_K 3x3 and P = (K.T)[:3,:] 3x4 passed BY
 * VALUE from host pointers; outputs pix [H,W,2] (normalised as the reference returns them) and z [H,W]. */
int mpf_backproject_project(const float *d_depth, const float *h_inv_k9, const float *h_P12, int H, int W,
                            float *d_pix, float *d_z, void *stream);
/* The two halves on their own, for the class-level drop-ins: BackprojectDepth.forward -> cam points [4,N] (rows X,Y,Z,1)
 * and Project3D.forward on arbitrary homogeneous points [4,N] with eps (geometry.py:55, :70). */
int mpf_backproject(const float *d_depth, const float *h_inv_k9, int H, int W, float *d_cam_points, void *stream);
int mpf_project3d(const float *d_points_4N, const float *h_P12, float eps, int H, int W, float *d_pix, float *d_z,
                  void *stream);
/* moving_obj.py:108-124, :153: select object/static projection by instance mask, to pixel units, truncate + clamp.
 * outputs p1 [H,W,2], z1 [H,W], safe_x/safe_y int64 [H,W], flow01 [H,W,2] */
int mpf_select_truncate(const float *d_p_static, const float *d_z_static, const float *d_p_obj, const float *d_z_obj,
                        const float *d_inst, int H, int W, float *d_p1, float *d_z1, int64_t *d_safe_x,
                        int64_t *d_safe_y, float *d_flow01, void *stream);
/* moving_obj.py:29-124 and :153 fused into one pass: depth from disparity, back-projection, the static and the object
 * projection (selected per pixel by the instance mask), pixel units, truncation + clamp, flow = p1 - p0.
 * inv_K 3x3, P_static = (K.T1)[:3,:], P_obj = (K.Ti)[:3,:] by value from host pointers. */
int mpf_moving_object_project(const float *d_disp, const float *h_inv_k9, const float *h_P_static12, const float *h_P_obj12,
                              const float *d_inst, int H, int W, float *d_p1, float *d_z1, int64_t *d_safe_x,
                              int64_t *d_safe_y, float *d_flow01, void *stream);
/* Order-preserving parallel equivalent of warping.c:6-33 on device buffers.  d_warped u8 [h,w,5] is fully written
 * (no need to zero it).  d_workspace: mpf_forward_warp_workspace(h,w) bytes of scratch. */
size_t mpf_forward_warp_workspace(int h, int w);
int mpf_forward_warp(const uint8_t *d_src, const int64_t *d_idx, const int64_t *d_idy, const float *d_z,
                     uint8_t *d_warped, int h, int w, void *d_workspace, size_t workspace_bytes, void *stream);
/* moving_obj.py:133-150: masks H, M, M' = dilate3x3(M), P = (M' == M), H' = H*P, each u8 [H,W] */
int mpf_warp_masks(const uint8_t *d_warped, int H, int W, uint8_t *d_Hm, uint8_t *d_M, uint8_t *d_Md, uint8_t *d_P,
                   uint8_t *d_Hp, void *stream);

/* moving_obj.py:29-150 in ONE call on device buffers: mpf_moving_object_project (fused into the first pass of the forward warp's sort:
 * the int64 targets are written for the caller but never read back), mpf_forward_warp, mpf_warp_masks - 5 launches, no allocation.
 * The source frame that is splatted (moving_obj.py:124), ONE of: d_src_u8 u8 [H,W,3], or d_src_f32_3HW float [3,H,W] in 0..1 whose
 * uint8 BGR form (utils/utils.py:174-177: rint(255 v), clamped - what Stage A+C writes as the pair's source frame) is splatted, converted
 * per winner on the fly, so that the chain needs no output of the render path.  Masks: all five pointers or none.
 * d_workspace: mpf_forward_warp_workspace(H, W) bytes, 256-byte aligned. */
typedef struct MpfMovingObjectOut {
    float *d_p1, *d_z1;               /* [H,W,2], [H,W] */
    int64_t *d_safe_x, *d_safe_y;     /* [H,W] */
    float *d_flow01;                  /* [H,W,2] */
    uint8_t *d_warped;                /* [H,W,5] */
    uint8_t *d_Hm, *d_M, *d_Md, *d_P, *d_Hp;   /* [H,W] each */
} MpfMovingObjectOut;
int mpf_moving_object_chain(const float *d_disp, const float *h_inv_k9, const float *h_P_static12, const float *h_P_obj12,
                            const float *d_inst, const uint8_t *d_src_u8, const float *d_src_f32_3HW, int H, int W,
                            const MpfMovingObjectOut *out, void *d_workspace, size_t workspace_bytes, void *stream);

/* ================= MPI producer network: 3x3 convolution engine ====================== */

/* One launch = one 3x3 / pad 1 / stride 1|2 convolution over S plane-images with its surrounding plumbing fused:
 * the LOADER synthesises the layer's (virtual) NHWC input, the EPILOGUE applies bias / BatchNorm / activation / gate.
 * Replaces, for S planes at once: ConvBNReLU (model/CPN/unet.py:6-15), GatedConv + ELU + BatchNorm
 * (model/CPN/decoder.py:10-71) and the expand / cat / ReflectionPad2d / upsample tensors around them
 * (model/CPN/unet.py:44-66, model/CPN/decoder.py:131-163).  Activations: fp16 NHWC, channels padded to a multiple of 8.
 * fp16 MFMA, fp32 accumulation, fp32 epilogue - the precision of the reference's own GPU run (.half(),
 * gen_3dphoto_dynamic_v2.py:46,59,82-84). */
#define MPF_CONV_LD_FMN_INPUT     0   /* (r,g,b,disparity,plane disparity,0,0,0): srcA = image f32 [3,H,W], srcB = disparity f32 [H,W], plane_vals f32 [S] */
#define MPF_CONV_LD_DIRECT        1   /* srcA f16 [S,Hin,Win,CA] */
#define MPF_CONV_LD_BILINEAR_CAT  2   /* x2 bilinear (align_corners) of srcA f16 [S,HA,WA,CA]  ++  srcB f16 [S,Hin,Win,CB]; fparams = {(HA-1)/(Hin-1), (WA-1)/(Win-1)} */
#define MPF_CONV_LD_NEAREST_PLANE 3   /* x2 nearest (or same size) of srcA f16 [S,HA,WA,CA] (CA may be 0)  ++  per-plane skip: srcB f16 [Hin,Win,CB-8] shared
                                         features * cm[s], then (cm[s], fm[s], 0 x6); cm, fm f32 [S,Hin,Win]; CB == 0: no skip */
#define MPF_CONV_LD_FMN_SYNTH      4   /* the feature-mask network's first layer never materialised: channels = relu(A' + plane_vals[s] * B'), srcA = A', srcB = B',
                                         both f32 [Hin,Win,16] (A' = its pre-activation output for plane value 0, B' = the plane channel's share) */
#define MPF_CONV_LD_BILINEAR_SYNTH 5   /* LD_BILINEAR_CAT whose skip source is synthesised the same way: srcB = A', cm = B' f32 [Hin,Win,16], CB = 16 */
#define MPF_CONV_LD_NEAREST_PHASE   6   /* LD_NEAREST_PLANE with HA = Hin / 2, reflection padding, PHASE-DECOMPOSED: on the upsampled source the 3x3 window of an output pixel covers
                                            2 x 2 distinct srcA pixels, chosen - with host-summed weights - by the pixel's phase (y & 1, x & 1); 4 taps instead of 9 there, the skip
                                            source as in LD_NEAREST_PLANE.  Chunks: ceil(CA / ct) of srcA, then ceil(CB / ct) of the skip; wpack = [chunkA][phase 2 py + px][ksteps(4 taps)]
                                            [nblk][64][8] ++ [chunkB][ksteps(9 taps)][nblk][64][8]  (engine.py: pack_weights_up).  Gated epilogues 2 / 6 only */
#define MPF_CONV_EP_AFFINE_RELU      0   /* out f16 [S,Hout,Wout,Cst] = relu(acc * ep[0][row] + ep[1][row]) */
#define MPF_CONV_EP_AFFINE_RELU_F32  1   /* same, output channel 0 only, out f32 [S,Hout,Wout] */
#define MPF_CONV_EP_GATED_ELU        2   /* g = (accF + ep[0][rowF]) * sigmoid(accM + ep[0][rowM]); out f16 NHWC = elu(g * ep[1][rowF] + ep[2][rowF]) */
#define MPF_CONV_EP_AFFINE_F32_NHWC  4   /* out f32 [S,Hout,Wout,Cst] = acc * ep[0][row] + ep[1][row], NO activation */
#define MPF_CONV_EP_GATED_PLANAR_F32 3   /* out f32 [S,Cst,Hout,Wout] = g (no BatchNorm / activation: the decoder's raw output layer) */
#define MPF_CONV_EP_GATED_ELU_PAIRED 6   /* EP_GATED_ELU with feature / gate rows interleaved (packed row 2c = feature c, 2c+1 = gate c; ep[1], ep[2] indexed by channel):
                                           24 output channels in 3 blocks instead of 4 */
#define MPF_CONV_EP_GATED_PLANAR_F32_PAIRED 5 /* the same from ONE 16-row block (nblk = 1, Cst <= 8): packed row 2c = feature c, row 2c+1 = gate c */

typedef struct MpfConvArgs {
    const void *srcA, *srcB;          /* see the loader */
    const float *cm, *fm;             /* per-plane masks at the conv-input resolution (LD_NEAREST_PLANE with CB > 0) */
    const float *plane_vals;          /* LD_FMN_INPUT */
    const void *wpack;                /* f16 weights in MFMA-fragment order: [nchunk][ksteps][nblk][64 lanes][8]   (mpiflow_amd/model/engine.py: pack_weights) */
    const float *ep;                  /* f32 [3][nblk*16] epilogue rows, in packed row order */
    void *out;
    int S, Hin, Win, Hout, Wout;      /* Hin x Win: the virtual conv input (after upsampling / concatenation) */
    int CA, CB, HA, WA;               /* padded channels of the two sources; size of source A */
    int ct, nchunk;                   /* channels staged per tap and chunk (8, 16, 32); number of chunks */
    int nblk, ncg;                    /* 16-row output blocks in total; workgroup column groups (nblk % ncg == 0) */
    int Cst;                          /* channels of the output tensor (NHWC pitch, or planes for the planar epilogue) */
    int loader, epi, stride, pad_mode; /* pad_mode 0 zero, 1 reflection */
    float fparams[4];
    int wlds;                         /* 1: the A fragments of a chunk are staged in LDS once per workgroup (many-chunk / many-block
                                         layers), 0: every wave loads its fragments from global memory (tuning choice, same results) */
    int plane_major;                  /* 1: the plane index is the fastest grid dimension (the S workgroups of a tile back to back: per-image sources
                                         shared by the planes stay in L2); scheduling only, same results */
    int bprime_table;                 /* LD_FMN_SYNTH / LD_BILINEAR_SYNTH: 1 = B' (srcB / cm) is the [3][3][16] table of its border classes (top / inner / bottom row x
                                         left / inner / right column: B' depends on the pixel only through which taps fall inside the image), 0 = an [Hin,Win,16] map */
    int pw;                           /* planes per workgroup (0 / 1: one): a workgroup walks pw consecutive planes at its tile position and computes what depends
                                         on the pixel only once (nblk / ncg <= 2 only; S % pw == 0); scheduling only, same results */
} MpfConvArgs;

int mpf_conv3x3_f16(const MpfConvArgs *args, void *stream);

/* The plane masks of the decoder in one pass over the feature-mask logits [S,H,W] (model/CPN/unet.py:68-69 softmax over the
 * planes; model/CPN/decoder.py:126-130 cumulative and context masks; :131-150 their adaptive_avg_pool2d at every decoder
 * scale).  Outputs: d_feature_mask [S,H,W] (optional), d_cum_mask [S,H,W], and for the five scales k = 2,4,8,16,32
 * d_cm[i], d_fm[i] [S,H/k,W/k] = the k x k block means of the context mask (1 - cumulative mask of the planes in front)
 * and of the feature mask.  d_cm / d_fm are HOST arrays of 5 device pointers.  H, W multiples of 32. */
int mpf_plane_masks(const float *d_logits, int S, int H, int W, float *d_feature_mask, float *d_cum_mask, float *const *d_cm,
                    float *const *d_fm, void *stream);

/* ---- the single-image part of the producer in fp32: RGBD ResNet-18 encoder + the decoder's bottleneck ----------------------
 * (model/CPN/encoder.py:20-101: conv1 7x7/2 + bn1 + relu, maxpool, layer1-4 of two BasicBlocks each;
 *  model/CPN/decoder.py:85-88,131-138: maxpool, conv1x1 + BN + LeakyReLU(0.1), maxpool, conv3x3, x2 nearest, conv3x3, x2 nearest, conv1x1.)
 * Activations: fp32 NHWC.  fp32 MFMA (v_mfma_f32_16x16x4_f32), fp32 epilogue. */

/* x = cat((image - mean) / std, disparity) (model/CPN/encoder.py:84-85,89-93): image f32 [3,H,W], disparity f32 [H,W] -> f32 [H,W,4] */
int mpf_encoder_input(const float *d_image_3HW, const float *d_disp_HW, int H, int W, float *d_out_HW4, void *stream);

/* One convolution (no bias) + per-channel affine (BatchNorm folded) [+ residual] + activation over one NHWC fp32 image:
 *   out[p, c] = act(conv(src)[p, c] * scale[c] + shift[c] (+ residual[p, c])),  zero padding.
 * up = 1: the convolution's input is the x2 nearest-neighbour upsampling of src (Hin x Win is the UPSAMPLED size; src is [Hin/2, Win/2, Cin]).
 * wpack: f32 [Cout/16][nsteps][64 lanes][4], nsteps = ceil(ksize^2 * Cin/4 / 4): lane (m = l % 16, g = l / 16) of step s holds, for j = 0..3,
 *   W[16 blk + m][channel 4 (v % (Cin/4)) + j][tap v / (Cin/4)] with v = 4 s + g, zero when the tap index exceeds ksize^2 - 1
 *   (mpiflow_amd/model/engine.py: pack_weights_f32).  Cin a power of two >= 4, Cout a multiple of 32.
 * out (f32) and out_f16 (f16, same NHWC shape: what the per-plane decoder's loaders read) are both optional, at least one is required. */
typedef struct MpfConv2dArgs {
    const float *src;
    const float *wpack;
    const float *scale, *shift;       /* f32 [Cout] */
    const float *residual;            /* f32 [Hout,Wout,Cout] or NULL */
    float *out;                       /* f32 [Hout,Wout,Cout] or NULL */
    void *out_f16;                    /* f16 [Hout,Wout,Cout] or NULL */
    int Hin, Win, Cin, Hout, Wout, Cout;
    int ksize, stride, pad;           /* ksize 1, 3 or 7; stride 1 or 2 */
    int up;                           /* 0, or 1: x2 nearest upsampling of src in front of the convolution */
    int act;                          /* 0 none, 1 ReLU, 2 LeakyReLU(slope) */
    float slope;
} MpfConv2dArgs;
int mpf_conv2d_f32(const MpfConv2dArgs *args, void *stream);

/* nn.MaxPool2d(3, stride 2, padding 1) on an NHWC fp32 image: [Hin,Win,C] -> [(Hin-1)/2+1, (Win-1)/2+1, C]; C a multiple of 4 */
int mpf_maxpool3x3s2_f32(const float *d_src_HWC, int Hin, int Win, int C, float *d_out, void *stream);

/* ---- the producer network's PARITY-GRADE engine: fp32 or fp64 throughout (mpiflow_amd/csrc/mpf_pconv.hip) -------------------------
 * Every convolution of MPIPredictor.forward (model/AdaMPI.py:55-78) in the arithmetic of the reference's CPU path: fp32 storage, fp32
 * products, fp32 accumulation (v_mfma_f32_16x16x4_f32), or fp64 throughout (v_mfma_f64_16x16x4_f64) - `dtype` selects.  Activations are
 * NHWC tensors of `dtype` with the channel count zero-padded to a multiple of 4; the accuracy mode behind
 * `gen_3dphoto_dynamic.py --model-engine hip --model-dtype fp32|fp64` (mpiflow_amd/model/precise.py: PrecisePredictor). */
#define MPF_DTYPE_F32 0
#define MPF_DTYPE_F64 1
#define MPF_DTYPE_F32X3      2 /* mpf_pconv only: fp32 tensors; every product a b from the three bf16 pieces each factor is exactly the sum of (six of the nine
                                * piece products: a relative 2^-24 per product dropped) on v_mfma_f32_16x16x32_bf16; accumulation as MPF_DTYPE_F32.  1 x 1 and 3 x 3.
                                * wpack: [nblk][steps][3 pieces][64 lanes][8] bf16, a step = two K-steps of the fp32 packing, every source padded to an even count */
#define MPF_DTYPE_F32X3_TILE 3 /* the same arithmetic for 3 x 3 / stride 1 / padding 1 layers with CA + CB <= 56 and nblk <= 3: the input tile of both sources is
                                * split once into LDS.  wpack: K-vector 4 t + g = (tap, 8-channel vector of the CONCATENATED channels, zero-padded to 8) */
#define MPF_DTYPE_F32X3_CHUNK 4 /* the same arithmetic for 3 x 3 / stride 1 / padding 1 layers of any width: per 32-channel chunk of the concatenated sources the input tile is
                                * split once into LDS, nine steps (one per tap) per chunk.  wpack: step = chunk * 9 + tap, K-vector g = 8-channel vector g of the chunk */
#define MPF_PCONV_EP_AFFINE       0   /* out [S,Hout,Wout,Cst] = act(acc * scale[row] + shift[row] (+ residual))   (ConvBNReLU model/CPN/unet.py:6-15; the encoder's conv + BN) */
#define MPF_PCONV_EP_AFFINE_MAP   1   /* same, row 0 only, out [S,Hout,Wout]   (the feature-mask logits, model/CPN/unet.py:66) */
#define MPF_PCONV_EP_GATED        2   /* g = accF * sigmoid(accM) (biases = initial accumulators); out NHWC = elu(g * scale[c] + shift[c])   (model/CPN/decoder.py:10-71) */
#define MPF_PCONV_EP_GATED_PLANAR 3   /* out [S,Cst,Hout,Wout] = g   (the decoder's raw output layer, model/CPN/decoder.py:164-165) */

/* One convolution over S plane-images (or one image, S = 1):  input = cat(A', B) along channels, A' = srcA or its x2 nearest up-sampling
 * (up = 1; HA = Hin / 2), zero or reflection padding, ksize 1 | 3 | 7, stride 1 | 2.
 * wpack: [nblk][nsteps][64 lanes][4] of dtype; K runs over source A's (tap, 4-channel vector) pairs, then over source B's: nsteps = nstA + nstB,
 *   nstX = ceil(ksize^2 * (CX/4) / 4).  Lane (m = l % 16, g = l / 16) of step s of source X holds, for j = 0..3, W[physical row 16 blk + m][channel
 *   4 (v % VX) + j of X][tap v / VX] with v = 4 s + g, VX = CX/4, zero past X's last tap.
 *   LOGICAL row L of a block (what the epilogue rows are indexed by) sits at physical row L for fp32 and at (L >> 2) + 4 (L & 3) for fp64
 *   (the C/D register layouts of the two MFMA instructions differ).  Gated epilogues: logical rows (2c, 2c+1) of the packed row sequence
 *   = (feature, gate) of channel c; bias [nblk*16] by logical row; scale / shift [nblk*8] by channel.  Affine epilogues: scale / shift
 *   [nblk*16] by logical row (conv bias folded into shift). */
typedef struct MpfPConvArgs {
    const void *srcA;                 /* dtype [S (or 1 when shareA), HA, WA, CA] */
    const void *srcB;                 /* dtype [S (or 1 when shareB), Hin, Win, CB] or NULL (CB = 0) */
    const void *wpack;
    const void *scale, *shift;        /* dtype, see above */
    const void *bias;                 /* dtype [nblk*16], gated epilogues only */
    const void *residual;             /* dtype [S,Hout,Wout,Cst] or NULL (EP_AFFINE) */
    void *out;
    int dtype;                        /* MPF_DTYPE_F32 | MPF_DTYPE_F64 | MPF_DTYPE_F32X3 | MPF_DTYPE_F32X3_TILE (tensors fp32 for the last two) */
    int S, Hin, Win, Hout, Wout;      /* Hin x Win: the virtual conv input (after up-sampling) */
    int HA, WA, CA, CB;
    int up, shareA, shareB;           /* up: 0 | 1 = x2 nearest up-sampling of srcA in front of the convolution | 2 (round 6; MPF_DTYPE_F32X3_TILE, 3 x 3 / stride 1 / reflection padding,
                                         CB = 0) = the same layer PHASE-DECOMPOSED: four 2 x 2 convolutions on the low-resolution map, wpack = [phase 2 py + px][row block][steps of
                                         4 taps] with the nine weights summed per phase in float64 (mpiflow_amd/model/precise.py: pack_weights_x3_tile_phase) */
    int ksize, stride, pad, pad_mode; /* pad_mode 0 zero, 1 reflection (pad 1) */
    int nblk, Cst, epi, act;          /* act: 0 none, 1 ReLU, 2 LeakyReLU(slope) (affine epilogues) */
    double slope;                     /* rounded to `dtype` by the kernel, as torch rounds the Python float to the tensor's dtype */
} MpfPConvArgs;
int mpf_pconv(const MpfPConvArgs *args, void *stream);

/* the tensors the reference builds with expand / cat / Upsample / adaptive_avg_pool2d, materialised in `dtype`:
 * mpf_pfmn_input      cat(image, disparity, plane disparity) per plane (model/CPN/unet.py:44-50) -> [S,H,W,8] (channels 5..7 zero)
 * mpf_pencoder_input  cat((image - mean) / std, disparity) (model/CPN/encoder.py:84-85,89-93) -> [H,W,4]
 * mpf_pbilinear2x     nn.Upsample(x2, bilinear, align_corners=True) (model/CPN/unet.py:42): [S,h,w,C] -> [S,2h,2w,C]
 * mpf_pper_plane      cat(feat * context_mask, context_mask, feature_mask) (model/CPN/decoder.py:140-150): feat [h,w,C], masks [S,h,w] -> [S,h,w,C+4]
 * mpf_pplane_masks    softmax over the planes (model/CPN/unet.py:68-69), cumulative / context masks (model/CPN/decoder.py:126-130) and their
 *                     k x k block means for k = 2..32 (adaptive_avg_pool2d, :143-146); d_cm / d_fm are HOST arrays of 5 device pointers
 * mpf_pmaxpool3x3s2   nn.MaxPool2d(3, 2, 1) on [Hin,Win,C] */
int mpf_pfmn_input(const float *d_image_3HW, const float *d_disp_HW, const float *d_plane_vals, int S, int H, int W, void *d_out, int dtype, void *stream);
int mpf_pencoder_input(const float *d_image_3HW, const float *d_disp_HW, int H, int W, void *d_out, int dtype, void *stream);
int mpf_pbilinear2x(const void *d_src, int S, int h, int w, int C, void *d_dst, int dtype, void *stream);
int mpf_pper_plane(const void *d_feat_hwC, const void *d_cm, const void *d_fm, int S, int h, int w, int C, void *d_out, int dtype, void *stream);
int mpf_pplane_masks(const void *d_logits, int S, int H, int W, void *d_feature_mask, void *d_cum_mask, void *d_context_mask, void *const *d_cm,
                     void *const *d_fm, int dtype, void *stream);
int mpf_pmaxpool3x3s2(const void *d_src_HWC, int Hin, int Win, int C, void *d_out, int dtype, void *stream);

/* THE REFERENCE'S FFI SYMBOL (external/forward_warping/warping.c:6; bound at moving_obj.py:12-13, called at :127-129).
 * Same name, same argument meaning, HOST pointers: src u8 [h*w*3], idx/idy int64 [h*w] (pre-clamped by the caller, as
 * in the reference), z f32 [h*w], warped u8 [h*w*5] (caller-owned).  Synchronous.  Runs the HIP kernels above on the
 * current device (copies in, warps, copies out); there is no CPU implementation behind it - without a GPU it prints
 * the HIP error to stderr and leaves `warped` untouched (the reference signature has no error channel). */
void forward_warping(const void *src, const void *idx, const void *idy, const void *z, void *warped, int h, int w);
/* same, with an error code */
int mpf_forward_warping_host(const void *src, const void *idx, const void *idy, const void *z, void *warped, int h,
                             int w);

#ifdef __cplusplus
}
#endif
#endif
