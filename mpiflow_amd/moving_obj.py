"""Drop-in for the reference's moving_obj.py: depth -> flow for an independently moving instance, forward warp,
collision / validity masks (Depthstillation heritage; reference moving_obj.py:16-168).

The reference's function returns None and its only product is a debug PNG (temp/res-%06d.png).  This version keeps
the name, the positional signature AND that behaviour - called as the reference calls it, it writes that PNG (the
source frame, the inpainted and the raw forward-warped frame and the validity mask stacked as at moving_obj.py:164-168;
the fifth panel, a colour-wheel rendering of the flow from the reference's visualisation module, is left out) and
returns None - while every per-pixel step runs in HIP (mpf_moving_object_project, mpf_forward_warp - byte-identical to
the C routine - and mpf_warp_masks); `return_intermediates=True` hands the device tensors back instead (tests, and
callers that want the forward-warped frame, masks and flow rather than a picture of them).
"""
import math
import random

import numpy as np
import torch

from . import host_math, ops


def object_pose(rng=None):
    """The object's rigid motion as the reference draws it (moving_obj.py:81-98): t = (U[.05,.1], -U[.05,.1], U[.05,.1]);
    three angles are drawn (and their signs) but then overwritten with zeros (:94), so only the RNG stream advances."""
    rng = rng or random
    sign = -1
    cix = (rng.random() * 0.05 + 0.05)
    ciy = -1 * (rng.random() * 0.05 + 0.05)
    ciz = (rng.random() * 0.05 + 0.05)
    for _ in range(3):
        _ = (rng.random() * math.pi / 72.0 + math.pi / 72.0) * (sign * (-1) ** rng.randrange(2))
    ai = torch.from_numpy(np.array([[[0, 0, 0]]], dtype=np.float32))
    tri = torch.from_numpy(np.array([[[cix, ciy, ciz]]][0])).float()
    return host_math.transformation_from_parameters(ai, tri)


def moveing_object_with_mask(depth_path, disp, rgb, K, inv_K, instance_mask, i, T_obj=None, write_debug_png=True,
                             inpaint="auto", return_intermediates=False):
    """(sic) reference moving_obj.py:16-168.

    :param disp: [1,1,h,w] disparity tensor;  :param rgb: [h,w,3] numpy holding 0..255;  :param K, inv_K: [3,3]
    :param instance_mask: [1,1,h,w] tensor (> 0 = the moving instance);  :param i: index for the debug file name
    :param T_obj: optional [1,4,4] object pose; default = drawn from `random` exactly as the reference does
    :param write_debug_png: write temp/res-%06d.png as the reference does (moving_obj.py:164-168); its only product
    :return: None, like the reference; with return_intermediates=True dict(p1, z1, safe_x, safe_y, flow_01, warped,
             masks{H,M,M',P,H'}, im1_raw, im1) - device tensors"""
    h, w = rgb.shape[:2]
    dev = disp.device if disp.is_cuda else torch.device("cuda")
    disp_d = disp.to(dev, torch.float32).reshape(h, w)
    K = host_math._cpu32(K).reshape(3, 3)
    inv_K = host_math._cpu32(inv_K).reshape(3, 3)
    K4 = torch.zeros((1, 4, 4)); K4[0, -1, -1] = 1.0; K4[:, :3, :3] = K          # :49-52
    T1 = host_math.transformation_from_parameters(torch.zeros(1, 1, 3), torch.zeros(1, 3))   # :43-47
    if T_obj is None:
        T_obj = object_pose()
    Ti = host_math._cpu32(T_obj).reshape(1, 4, 4)
    P1 = torch.matmul(K4, T1)[:, :3, :][0]                                       # geometry.py:65
    Pi = torch.matmul(K4, Ti)[:, :3, :][0]
    inst = instance_mask.to(dev, torch.float32).reshape(h, w)
    # :29-30 depth, :63-66 / :101-105 the two projections, :108-124 select + truncate, :153 flow - one fused kernel
    p1, z1, safe_x, safe_y, flow_01 = ops.moving_object_project(disp_d, inv_K, P1, Pi, inst)
    img = torch.from_numpy(np.ascontiguousarray(rgb)).to(dev).float().reshape(-1).to(torch.uint8)   # :20, :124
    warped = ops.forward_warp(img, safe_x, safe_y, z1, h, w)                     # :127-129
    masks = ops.warp_masks(warped)                                               # :133-150
    im1_raw = warped[:, :, 0:3]
    hole = (1 - masks["H"]).to(torch.uint8)
    from .utils.utils import _inpaint                                            # :162 cv2.inpaint(im1_raw, 1 - H, 3, INPAINT_TELEA)
    im1 = torch.from_numpy(np.ascontiguousarray(_inpaint(im1_raw.contiguous(), hole, inpaint, algo="telea"))).to(dev)
    out = dict(p1=p1, z1=z1, safe_x=safe_x, safe_y=safe_y, flow_01=flow_01, warped=warped, masks=masks, im1_raw=im1_raw, im1=im1)
    if write_debug_png:
        import os
        from PIL import Image
        os.makedirs("temp", exist_ok=True)
        m3 = (masks["H"] * 255).unsqueeze(-1).repeat(1, 1, 3)
        res = np.vstack([np.asarray(rgb).astype(np.uint8), im1.cpu().numpy(), im1_raw.cpu().numpy(), m3.cpu().numpy()])
        Image.fromarray(res[:, :, ::-1].copy()).save("temp/res-{:06d}.png".format(i))
    return out if return_intermediates else None
