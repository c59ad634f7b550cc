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


def projection_matrices(K, T_obj):
    """P_static = (K4 . T1)[:3], P_obj = (K4 . Ti)[:3] with the reference's expressions (moving_obj.py:43-52, geometry.py:65)."""
    K = host_math._cpu32(K).reshape(3, 3)
    K4 = torch.zeros((1, 4, 4)); K4[0, -1, -1] = 1.0; K4[:, :3, :3] = K          # :49-52
    T1 = host_math.transformation_from_parameters(torch.zeros(1, 1, 3), torch.zeros(1, 3))   # :43-47
    Ti = host_math._cpu32(T_obj).reshape(1, 4, 4)
    return torch.matmul(K4, T1)[:, :3, :][0].contiguous(), torch.matmul(K4, Ti)[:, :3, :][0].contiguous()


class MovingObjectChain:
    """moving_obj.py:29-150 for a STREAM of frames of one size: matrices prepared once per object pose, `n_buffers` preallocated
    output sets used round-robin (a set is rewritten n_buffers run() calls later), one C call = 3 launches per frame, no
    allocation.  pipeline.OverlappedPairRenderer runs it on a side stream underneath the pair launches."""

    def __init__(self, H, W, K, inv_K, device, T_obj=None, n_buffers=2):
        self.H, self.W, self.device = H, W, torch.device(device)
        self.inv_K = host_math._cpu32(inv_K).reshape(9).contiguous()
        self.set_object_pose(K, object_pose() if T_obj is None else T_obj)
        self.bufs = [ops.MovingObjectBuffers(H, W, self.device) for _ in range(n_buffers)]
        self._next = 0

    def set_object_pose(self, K, T_obj):
        self.P_static, self.P_obj = projection_matrices(K, T_obj)

    def run(self, disp, inst, src_u8, which=None):
        """disp [H,W] f32, inst [H,W] f32 (> 0 = the moving instance), src_u8 [H,W,3] - device tensors; launches on the current
        stream into output set `which` (default: round-robin).  -> ops.MovingObjectBuffers (p1, z1, safe_x, safe_y, flow_01, warped, masks)"""
        if which is None:
            which, self._next = self._next, (self._next + 1) % len(self.bufs)
        b = self.bufs[which]
        return ops.moving_object_chain(disp, self.inv_K, self.P_static, self.P_obj, inst, src_u8, bufs=b)


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
    inv_K = host_math._cpu32(inv_K).reshape(3, 3)
    if T_obj is None:
        T_obj = object_pose()
    P1, Pi = projection_matrices(K, T_obj)                                       # :43-52, geometry.py:65
    inst = instance_mask.to(dev, torch.float32).reshape(h, w)
    img = torch.from_numpy(np.ascontiguousarray(rgb)).to(dev).float().reshape(-1).to(torch.uint8)   # :20, :124
    # :29-30 depth, :63-66 / :101-105 the two projections, :108-124 select + truncate, :153 flow (fused into the sort's first pass),
    # :127-129 the forward splat, :133-150 the masks - one C call
    b = ops.moving_object_chain(disp_d, inv_K, P1, Pi, inst, img.reshape(h, w, 3))
    p1, z1, safe_x, safe_y, flow_01, warped, masks = b.p1, b.z1, b.safe_x, b.safe_y, b.flow_01, b.warped, b.masks
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
