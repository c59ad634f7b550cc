"""Drop-in for the reference's older per-image module "utils/utils copy.py" (a file name with a space, so the reference itself can only
load it by path): the `copy` pose constants (no `opt` argument, cz range 0.1, translations x0.3, angles x0.2) and, next
to the merged frame, the DEPTH-ORDERED frame of :295-303 - where both rendered layers cover a target pixel and the object layer lies
behind the background layer, the dynamic layer's pixel wins - with its own hole fill (:312-315).

The reference returns that frame only inside a visualisation montage (`res`, :316-324: eight panels plus flow colour wheels, written to
data_path/image.png); montages are out of scope here, so the fourth return value is a dict of the arrays the montage's
depth column is made of."""
import torch

from .. import host_math, pipeline
from . import utils as _v2
from .utils import (image_to_tensor, disparity_to_tensor, gen_swing_path, render_novel_view_dynamic,  # noqa: F401
                    transformation_from_parameters, HomographySample, mpi_rendering)


def generate_random_pose(base_motions=[0.05, 0.05, 0.05]):
    """"utils/utils copy.py":121-160 -> [4,4] tensor (CPU)"""
    return host_math.generate_random_pose(base_motions=base_motions, profile="copy")


def render_3dphoto_dynamic(src_imgs, obj_mask, disp, mpi_all_src, disparity_all_src, k_src, k_tgt, data_path=None, name=None,
                           hard_flow=False, mask_thresh=0.99, inpaint="auto"):
    """"utils/utils copy.py":164-326.  -> (flow_mix [H,W,2] f32, src_np [H,W,3] u8 BGR, inpainted [H,W,3] u8 BGR,
    dict(frame_mix_depth, frame_mix_depth_inpainted [H,W,3] u8 BGR, depth, depth_dync [H,W] f32, depth_mask [H,W] bool)) as numpy arrays."""
    name = name.split(".")[0]
    dev = mpi_all_src.device
    h, w = mpi_all_src.shape[-2:]
    cam_ext_dynamic = host_math.generate_random_pose(0.1, profile="copy")                                     # :210
    cam_ext = host_math.generate_random_pose(0.1, base_motions=[0, 0, 0], profile="copy")                     # :211
    out = pipeline.render_pair(src_imgs[0].to(dev, torch.float32), obj_mask.reshape(h, w).to(dev, torch.float32),
                               mpi_all_src[0].to(torch.float32), disparity_all_src[0], k_src, cam_ext, cam_ext_dynamic,
                               thresh=mask_thresh, hard_flow=hard_flow, depth_ordered=True)
    inpainted = _v2._inpaint(out["frame_mix"], out["fill_mask"], inpaint)                                       # :309-311
    depth_inpainted = _v2._inpaint(out["frame_mix_depth"], out["fill_mask"], inpaint)                           # :312-315 (fill_mask_depth = fill_mask.copy())
    res = dict(frame_mix_depth=out["frame_mix_depth"].cpu().numpy(), frame_mix_depth_inpainted=depth_inpainted,
               depth=out["view_cam"]["depth"].cpu().numpy(), depth_dync=out["view_dyn"]["depth"].cpu().numpy(),
               depth_mask=out["depth_mask"].cpu().numpy().astype(bool))
    return out["flow_mix"].cpu().numpy(), out["src_np"].cpu().numpy(), inpainted, res
