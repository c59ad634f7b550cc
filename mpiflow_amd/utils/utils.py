"""Drop-in for the reference's per-image pipeline module utils/utils.py (render_3dphoto_dynamic,
render_novel_view_dynamic, generate_random_pose, gen_swing_path, image/disparity loaders) on MI355X.

Same positional signatures and return values as the reference; tensors live on the GPU; arithmetic runs in the fused
HIP kernels of libmpiflow_hip.so (mpiflow_amd.pipeline).  Inputs of any float dtype are promoted to fp32 - the
parity target is the reference's fp32 CPU path.
"""
import math

import numpy as np
import torch

from .. import host_math, ops, pipeline
from ..geometry import transformation_from_parameters  # noqa: F401
from .mpi import mpi_rendering  # noqa: F401
from .mpi.homography_sampler import HomographySample  # noqa: F401


def image_to_tensor(img_path, unsqueeze=True):
    """RGB image file -> [1,3,h,w] float in [0,1]   (reference utils/utils.py:35-39; torchvision's ToTensor = /255)"""
    from PIL import Image
    rgb = torch.from_numpy(np.asarray(Image.open(img_path).convert("RGB")).copy()).permute(2, 0, 1).float().div(255)
    return rgb.unsqueeze(0) if unsqueeze else rgb


def disparity_to_tensor(disp_path, unsqueeze=True):
    """grey-scale disparity file -> [1,1,h,w] float in [0,1]   (reference :42-52; cv2.imread(path, 0) / 255)"""
    from ..io_formats import read_disparity_u8
    disp = read_disparity_u8(disp_path).astype(np.float64) / 255      # 16-bit files: high byte, as cv2 - not PIL's saturating convert("L")
    disp = torch.from_numpy(disp)[None, ...]
    if unsqueeze:
        disp = disp.unsqueeze(0)
    return disp.float()


def gen_swing_path(num_frames=90, r_x=0.14, r_y=0.0, r_z=0.10):
    """List of [4,4] poses on a swing path (reference :55-62)."""
    t = torch.arange(num_frames) / (num_frames - 1)
    poses = torch.eye(4).repeat(num_frames, 1, 1)
    poses[:, 0, 3] = r_x * torch.sin(2.0 * math.pi * t)
    poses[:, 1, 3] = r_y * torch.cos(2.0 * math.pi * t)
    poses[:, 2, 3] = r_z * (torch.cos(2.0 * math.pi * t) - 1.0)
    return poses.unbind()


def generate_random_pose(ext_cz, base_motions=[0.1, 0.1, 0.1]):
    """Random camera extrinsic from Python's global `random` stream (reference :121-156) -> [4,4] tensor (CPU; the
    pose only ever feeds host-side 3x3 algebra)."""
    return host_math.generate_random_pose(ext_cz, base_motions=base_motions)


def have_cv2():
    try:
        import cv2  # noqa: F401
        return True
    except Exception:
        return False


INPAINT_METHODS = ("auto", "cv2", "builtin", "ns", "telea", "peel", "hip", "none")


def resolve_inpaint(method):
    """auto -> 'cv2' when OpenCV is installed (the reference's own call), else 'builtin' (the same algorithm, restated)"""
    if method not in INPAINT_METHODS:
        raise ValueError("inpaint must be one of %s" % (INPAINT_METHODS,))
    if method == "auto":
        return "cv2" if have_cv2() else "builtin"
    return "peel" if method == "hip" else method


def _inpaint(frame_dev, hole_dev, method, algo="ns"):
    """Row A13.  `algo` is the algorithm of the reference line being replaced: 'ns' for
    cv2.inpaint(frame_mix, fill_mask, 3, cv2.INPAINT_NS) (utils/utils.py:284-286), 'telea' for
    cv2.inpaint(im1_raw, 1 - H, 3, cv2.INPAINT_TELEA) (moving_obj.py:162).  `method`:
      'cv2'      the reference's own call (third-party; needs OpenCV)
      'builtin'  that algorithm as restated in libmpiflow_hip.so (mpf_inpaint_host; on the host, like the reference's call);
                 'ns' / 'telea' force one of the two
      'peel' (alias 'hip')  the onion-peel GPU kernel - NOT OpenCV's algorithm, an explicit opt-in only
      'none'     leave the holes as they are."""
    method = resolve_inpaint(method)
    if method in ("ns", "telea"):
        method, algo = "builtin", method
    if method == "cv2":
        import cv2
        return cv2.inpaint(frame_dev.cpu().numpy(), hole_dev.cpu().numpy().astype(np.uint8), 3,
                           cv2.INPAINT_TELEA if algo == "telea" else cv2.INPAINT_NS)
    if method == "builtin":
        return ops.inpaint_host(frame_dev.cpu().numpy(), hole_dev.cpu().numpy(), 3, ops.INPAINT_TELEA if algo == "telea" else ops.INPAINT_NS)
    if method == "peel":
        return ops.fill_holes(frame_dev, hole_dev).cpu().numpy()
    return frame_dev.cpu().numpy()


def render_3dphoto_dynamic(opt, src_imgs, obj_mask, disp, mpi_all_src, disparity_all_src, k_src, k_tgt, data_path=None,
                           name=None, hard_flow=False, mask_thresh=0.99, inpaint="auto", return_intermediates=False, pose_profile="v2"):
    """One training pair from one MPI (reference utils/utils.py:159-288).

    Draws the dynamic pose then the camera pose from `random` (same order as the reference), blends the source image
    into the planes, renders the object layer with the camera pose and the background layer with the dynamic pose
    (sic), merges by the rendered masks and fills the holes.
    :return: (flow_mix [H,W,2] float32, src_np [H,W,3] uint8 BGR, inpainted [H,W,3] uint8 BGR, None) as numpy arrays
    """
    name = name.split(".")[0]
    dev = mpi_all_src.device
    S = mpi_all_src.shape[1]
    h, w = mpi_all_src.shape[-2:]
    ext_cz = opt.ext_cz if pose_profile == "v2" else 0.1
    cam_ext_dynamic = host_math.generate_random_pose(ext_cz, profile=pose_profile)
    cam_ext = host_math.generate_random_pose(ext_cz, base_motions=[0, 0, 0], profile=pose_profile)
    out = pipeline.render_pair(src_imgs[0].to(dev, torch.float32), obj_mask.reshape(h, w).to(dev, torch.float32),
                               mpi_all_src[0].to(torch.float32), disparity_all_src[0], k_src, cam_ext, cam_ext_dynamic,
                               thresh=mask_thresh, hard_flow=hard_flow)
    inpainted = _inpaint(out["frame_mix"], out["fill_mask"], inpaint)
    flow_mix = out["flow_mix"].cpu().numpy()
    src_np = out["src_np"].cpu().numpy()
    if return_intermediates:
        return flow_mix, src_np, inpainted, None, out
    return flow_mix, src_np, inpainted, None


def render_novel_view_dynamic(obj_mask, mpi_all_rgb_src, mpi_all_sigma_src, disparity_all_src, G_tgt_src, K_src_inv, K_tgt,
                              K_src, src_pose, homography_sampler, hard_flow=False):
    """One posed view of an already-blended MPI (reference utils/utils.py:291-349).
    :return: (tgt_imgs_syn [1,3,H,W], tgt_depth_syn [1,1,H,W], flow_syn [1,2,H,W] clipped to +-200, obj_mask [1,1,H,W])

    Fused path: the two channel-planar tensors are consumed IN PLACE - Stage B reads the three colour planes and the sigma plane where
    they lie (mpf_warp_composite_split; no concatenation, no repack: the reference builds an 8-channel copy of the stack per call,
    utils/mpi/mpi_rendering.py:288-301), xyz channels are evaluated analytically, and the flow comes from the source-frame weights of
    the sigma tensor alone (mpf_src_flow)."""
    B, S = disparity_all_src.size()
    assert B == 1, "the reference's entry point is batch-1 (utils/utils.py:314 indexes [0])"
    H, W = mpi_all_rgb_src.shape[-2:]
    dev = mpi_all_rgb_src.device
    d = host_math.plane_depths(disparity_all_src[0])
    H_ts, H_st = host_math.homographies(G_tgt_src, K_src_inv, K_tgt, d)
    rgb_S3HW, sigma_S1HW = mpi_all_rgb_src[0], mpi_all_sigma_src[0]
    quads = ops.mask_quads(obj_mask.reshape(H, W).to(dev, torch.float32), False)
    v = ops.warp_composite_split(rgb_S3HW, sigma_S1HW, quads, H_st, K_src_inv, G_tgt_src, d)
    if hard_flow:
        # :126-130 in one pass over the sigma tensor (mpf_src_flow_hard): the arg-max-weight plane's flow, nothing per-plane materialised
        flow = ops.src_flow_hard(sigma_S1HW.to(torch.float32), K_src_inv, d, H_ts.unsqueeze(0), flow_clip=200.0)[0:1]
    else:
        # source-frame weights: the Stage A+C kernel's flow-only body on the sigma tensor
        flow = ops.src_flow(sigma_S1HW, K_src_inv, d, H_ts.unsqueeze(0), flow_clip=200.0)[0:1]
    return v["rgb"].unsqueeze(0), v["depth"].reshape(1, 1, H, W), flow.reshape(1, 2, H, W), v["objmask"].reshape(1, 1, H, W)
