"""Drop-in for the on-path functions of the reference's utils/mpi/mpi_rendering.py - same names, argument order and
return tuples - with the per-pixel arithmetic in HIP kernels (libmpiflow_hip.so).

These are the *generic* forms: they accept and return the same materialised [B,S,C,H,W] tensors as the reference, so
each function is a drop-in on its own and reproduces the reference's semantics exactly (incl. sampling the xyz
channels rather than recomputing them).  The fused fast path that never materialises an [S,...] intermediate lives in
mpiflow_amd.pipeline / utils.utils.render_3dphoto_dynamic.

Also provided although MPI-Flow's generation never takes them: alpha_composition / use_alpha=True (:42-59) and
get_xyz_from_depth (:157-177).  Not provided: disparity_consistency_src_to_tgt (:180-210), a training loss of the MINE code
base this module was taken from.
"""
import torch

from ... import host_math, ops
from .homography_sampler import HomographySample
from .rendering_utils import transform_G_xyz  # noqa: F401  (re-exported like the reference module does)


def _flat(t):                      # [B,S,C,H,W] -> per-batch [S,C,H*W] views
    B, S, C, H, W = t.shape
    return t.reshape(B, S, C, H * W)


def render(rgb_BS3HW, sigma_BS1HW, xyz_BS3HW, src_sigma_BS1HW=None, src_flow_BS2HW=None, src_xyz_BS3HW=None,
           use_alpha=False, is_bg_depth_inf=False, hard_flow=False, obj_mask=None):
    """reference utils/mpi/mpi_rendering.py:7-39 -> (imgs_syn, depth_syn, blend_weights, weights, flowA2B, obj_mask)"""
    if use_alpha:
        # The reference's use_alpha branch (:33-38) computes alpha_composition results and then fails at its return statement
        # (`flowA2B` is only bound in the other branch): same exception here.  alpha_composition() itself is provided below.
        raise UnboundLocalError("local variable 'flowA2B' referenced before assignment (reference utils/mpi/mpi_rendering.py:39, use_alpha=True)")
    imgs_syn, depth_syn, blend_weights, weights, flowB2A, obj_mask = plane_volume_rendering(
        rgb_BS3HW, sigma_BS1HW, xyz_BS3HW, src_sigma_BS1HW, src_flow_BS2HW, src_xyz_BS3HW, is_bg_depth_inf,
        hard_flow=hard_flow, obj_mask=obj_mask)
    flowA2B = None
    if src_sigma_BS1HW is not None:
        flowA2B = plane_volume_rendering_flow(src_sigma_BS1HW, src_flow_BS2HW, src_xyz_BS3HW, is_bg_depth_inf,
                                              hard_flow=hard_flow)
    return imgs_syn, depth_syn, blend_weights, weights, flowA2B, obj_mask


def alpha_composition(alpha_BK1HW, value_BKCHW):
    """reference :42-59 ("Single-View View Synthesis with Multiplane Images") -> (value_composed BxCxHxW, weights BxKx1xHxW)"""
    B, K, _, H, W = alpha_BK1HW.size()
    C = value_BKCHW.size(2)
    vals, wts = [], []
    for b in range(B):
        r = ops.alpha_composite(alpha_BK1HW[b, :, 0], value_BKCHW[b])
        vals.append(r["out"].reshape(C, H, W))
        wts.append(r["weights"].reshape(K, 1, H, W))
    return torch.stack(vals), torch.stack(wts)


def get_xyz_from_depth(meshgrid_homo, depth, K_inv):
    """xyz = (K^-1 . (x,y,1)) * depth   (reference :157-177): meshgrid 3xHxW (only its size is read), depth Bx1xHxW,
    K_inv Bx3x3 -> Bx3xHxW"""
    H, W = meshgrid_homo.size(1), meshgrid_homo.size(2)
    B, _, H_d, W_d = depth.size()
    assert H == H_d and W == W_d
    return torch.stack([ops.backproject(depth[b, 0], K_inv[b])[:3].reshape(3, H, W) for b in range(B)])


def plane_volume_rendering(rgb_BS3HW, sigma_BS1HW, xyz_BS3HW, src_sigma_BS1HW, src_flow_BS2HW, src_xyz_BS3HW,
                           is_bg_depth_inf, hard_flow=False, obj_mask=None):
    """reference :62-99 -> (rgb_out Bx3xHxW, depth_out Bx1xHxW, transparency_acc BxSx1xHxW, weights BxSx1xHxW,
    flow_out Bx2xHxW | None, obj_mask Bx1xHxW | as passed)"""
    B, S, _, H, W = sigma_BS1HW.size()
    rgbs, depths, taccs, wts, flows, oms = [], [], [], [], [], []
    for b in range(B):
        extra = None
        if src_sigma_BS1HW is not None:
            parts = [src_flow_BS2HW[b].to(torch.float32)]
            if obj_mask is not None:
                parts.append(obj_mask[b].to(torch.float32))
            extra = torch.cat(parts, dim=1).contiguous()
        r = ops.volume_render(rgb_BS3HW[b], sigma_BS1HW[b, :, 0], xyz_BS3HW[b], extra_SEN=extra)
        depth = r["depth"]
        if is_bg_depth_inf:      # :146-149 (DTU option): sum(w z) + (1 - sum w) * 1000 instead of the normalised depth
            wsum = ops.weighted_sum(r["weights"])[0]
            wz = ops.weighted_sum(r["weights"], xyz_BS3HW[b][:, 2:3].contiguous())[0]
            depth = wz + (1 - wsum) * 1000
        rgbs.append(r["rgb"]); depths.append(depth.unsqueeze(0)); taccs.append(r["tacc"].unsqueeze(1)); wts.append(r["weights"].unsqueeze(1))
        if extra is not None:
            flows.append(r["extra"][0:2])
            if obj_mask is not None:
                oms.append(r["extra"][2:3])
    flow_out = torch.stack(flows) if flows else None
    om_out = torch.stack(oms) if oms else obj_mask
    return torch.stack(rgbs), torch.stack(depths), torch.stack(taccs), torch.stack(wts), flow_out, om_out


def plane_volume_rendering_flow(src_sigma_BS1HW, src_flow_BS2HW, src_xyz_BS3HW, is_bg_depth_inf, hard_flow=False):
    """reference :102-139 -> flow_out Bx2xHxW (hard_flow: the arg-max-weight plane's flow)"""
    B = src_sigma_BS1HW.size(0)
    out = []
    for b in range(B):
        r = ops.volume_render(None, src_sigma_BS1HW[b, :, 0], src_xyz_BS3HW[b], extra_SEN=src_flow_BS2HW[b], hard=hard_flow,
                              want_tacc=False, want_weights=False)
        out.append(r["extra"])
    return torch.stack(out)


def weighted_sum_mpi(rgb_BS3HW, xyz_BS3HW, weights, is_bg_depth_inf):
    """reference :142-154 -> (rgb_out Bx3xHxW, depth_out Bx1xHxW)"""
    B = rgb_BS3HW.size(0)
    rgbs, depths = [], []
    for b in range(B):
        w = weights[b, :, 0]
        wsum = ops.weighted_sum(w)
        rgbs.append(ops.weighted_sum(w, rgb_BS3HW[b]))
        wz = ops.weighted_sum(w, xyz_BS3HW[b][:, 2:3].contiguous())
        depths.append(wz + (1 - wsum) * 1000 if is_bg_depth_inf else wz / (wsum + 1e-5))
    return torch.stack(rgbs), torch.stack(depths)


def get_src_xyz_from_plane_disparity(meshgrid_src_homo, mpi_disparity_src, K_src_inv):
    """xyz_src[b,s] = (K^-1 . (x,y,1)) / disparity[b,s]   (reference :213-239)

    :param meshgrid_src_homo: 3xHxW - must be the standard (x, y, 1) grid of HomographySample (its values are
                              regenerated from pixel indices in the kernel; only H and W are read from it)
    :param mpi_disparity_src: BxS
    :param K_src_inv: Bx3x3
    :return: BxSx3xHxW"""
    B, S = mpi_disparity_src.size()
    H, W = meshgrid_src_homo.size(1), meshgrid_src_homo.size(2)
    dev = mpi_disparity_src.device if mpi_disparity_src.is_cuda else meshgrid_src_homo.device
    out = [ops.src_xyz(K_src_inv[b], host_math.plane_depths(mpi_disparity_src[b]), H, W, dev) for b in range(B)]
    return torch.stack(out)


def get_tgt_xyz_from_plane_disparity(xyz_src_BS3HW, G_tgt_src):
    """xyz_tgt = G . [xyz_src; 1]   (reference :242-256) -> BxSx3xHxW"""
    B, S, _, H, W = xyz_src_BS3HW.size()
    out = [ops.transform_xyz(G_tgt_src[b], xyz_src_BS3HW[b].reshape(S, 3, H * W)).reshape(S, 3, H, W) for b in range(B)]
    return torch.stack(out)


def render_tgt_rgb_depth(H_sampler: HomographySample, mpi_rgb_src, mpi_sigma_src, mpi_disparity_src, xyz_tgt_BS3HW,
                         xyz_src_BS3HW, G_tgt_src, K_src_inv, K_tgt, mpi_flow_src=None, use_alpha=False,
                         is_bg_depth_inf=False, hard_flow=False, obj_mask=None):
    """reference :259-349 -> (tgt_rgb_syn Bx3xHxW, tgt_depth_syn Bx1xHxW, tgt_mask Bx1xHxW, flowA2B Bx2xHxW,
    tgt_obj_mask_sync Bx1xHxW).  `mpi_flow_src` is accepted and ignored exactly as in the reference (:267)."""
    B, S, _, H, W = mpi_rgb_src.size()
    mpi_depth_src = torch.reciprocal(mpi_disparity_src.detach().to("cpu", torch.float32))          # :284
    mpi_xyz_src = torch.cat((mpi_rgb_src.to(torch.float32), mpi_sigma_src.to(torch.float32), xyz_tgt_BS3HW.to(torch.float32)), dim=2)
    if obj_mask is not None:
        mpi_xyz_src = torch.cat((mpi_xyz_src, obj_mask.to(torch.float32)), dim=2)
    G_Bs44 = G_tgt_src.unsqueeze(1).repeat(1, S, 1, 1).contiguous().reshape(B * S, 4, 4)
    Kinv_Bs33 = K_src_inv.unsqueeze(1).repeat(1, S, 1, 1).contiguous().reshape(B * S, 3, 3)
    Kt_Bs33 = K_tgt.unsqueeze(1).repeat(1, S, 1, 1).contiguous().reshape(B * S, 3, 3)
    tgt, tgt_mask_BsHW, _ = H_sampler.sample(mpi_xyz_src.view(B * S, -1, H, W), mpi_depth_src.view(B * S), G_Bs44, Kinv_Bs33, Kt_Bs33)
    flowB2A = H_sampler.sample_inverse(mpi_xyz_src.view(B * S, -1, H, W), mpi_depth_src.view(B * S), G_Bs44, Kinv_Bs33, Kt_Bs33)
    flowB2A = flowB2A.permute(0, 3, 1, 2).reshape(B, S, 2, H, W)                                     # :316
    tgt = tgt.view(B, S, -1, H, W)
    tgt_rgb, tgt_sigma, tgt_xyz = tgt[:, :, 0:3], tgt[:, :, 3:4], tgt[:, :, 4:7]
    tgt_om = tgt[:, :, 7:] if obj_mask is not None else None
    tgt_sigma = torch.where(tgt_xyz[:, :, -1:] >= 0, tgt_sigma, torch.zeros_like(tgt_sigma))        # :336-338
    rgb_syn, depth_syn, _, _, flowA2B, om_sync = render(tgt_rgb.contiguous(), tgt_sigma.contiguous(), tgt_xyz.contiguous(),
                                                        mpi_sigma_src, flowB2A, xyz_src_BS3HW, use_alpha=use_alpha,
                                                        is_bg_depth_inf=is_bg_depth_inf, hard_flow=hard_flow,
                                                        obj_mask=tgt_om.contiguous() if tgt_om is not None else None)
    tgt_mask = torch.sum(tgt_mask_BsHW.view(B, S, H, W).to(torch.float32), dim=1, keepdim=True)      # :347 (exact integers)
    return rgb_syn, depth_syn, tgt_mask, flowA2B, om_sync
