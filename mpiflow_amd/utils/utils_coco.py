"""Drop-in for the reference's utils/utils_coco.py: the per-image pipeline with the COCO pose constants (no `opt`
argument, cz range 0.1, angle scale 0.5).  Everything but the pose sampler is utils/utils.py."""
from .. import host_math
from . import utils as _v2
from .utils import (image_to_tensor, disparity_to_tensor, gen_swing_path, render_novel_view_dynamic,  # noqa: F401
                    transformation_from_parameters, HomographySample, mpi_rendering)


def generate_random_pose(base_motions=[0.1, 0.1, 0.1]):
    """utils/utils_coco.py:121-154 -> [4,4] tensor (CPU)"""
    return host_math.generate_random_pose(base_motions=base_motions, profile="coco")


def render_3dphoto_dynamic(src_imgs, obj_mask, disp, mpi_all_src, disparity_all_src, k_src, k_tgt, data_path=None, name=None,
                           hard_flow=False, mask_thresh=0.99, inpaint="auto"):
    """utils/utils_coco.py:157-250: same 4-tuple as utils.utils.render_3dphoto_dynamic, poses drawn with the COCO constants."""
    return _v2.render_3dphoto_dynamic(None, src_imgs, obj_mask, disp, mpi_all_src, disparity_all_src, k_src, k_tgt, data_path=data_path,
                                      name=name, hard_flow=hard_flow, mask_thresh=mask_thresh, inpaint=inpaint, pose_profile="coco")
