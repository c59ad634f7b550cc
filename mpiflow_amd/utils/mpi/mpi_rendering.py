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


# ---- provenance tags: which tensors are "the standard ones" ---------------------------------------------------------------------------
# render_tgt_rgb_depth receives xyz_src / xyz_tgt as materialised tensors.  When they are what get_src_xyz_from_plane_disparity /
# get_tgt_xyz_from_plane_disparity of THIS module returned for the same (K_src_inv, disparities, G_tgt_src) - the reference's only call site
# builds them exactly so (utils/utils.py:303-310) - the fused kernels, which evaluate those affine fields in registers, apply.  The tag is a
# Python attribute on the returned tensor: (what, bytes of the small host matrices it was built from, tensor version at tagging time).  A
# tensor that was modified in place, sliced, copied or built any other way carries no valid tag and takes the generic kernels.

def _key(*small):
    return tuple(host_math._cpu32(t).contiguous().numpy().tobytes() for t in small)


def _tag(t, what, key):
    t._mpf_std = (what, key, t._version)
    return t


def _tagged(t, what):
    tag = getattr(t, "_mpf_std", None)
    return tag[1] if tag is not None and tag[0] == what and tag[2] == t._version else None


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
    return _tag(torch.stack(out), "src", _key(K_src_inv, mpi_disparity_src))


def get_tgt_xyz_from_plane_disparity(xyz_src_BS3HW, G_tgt_src):
    """xyz_tgt = G . [xyz_src; 1]   (reference :242-256) -> BxSx3xHxW"""
    B, S, _, H, W = xyz_src_BS3HW.size()
    out = [ops.transform_xyz(G_tgt_src[b], xyz_src_BS3HW[b].reshape(S, 3, H * W)).reshape(S, 3, H, W) for b in range(B)]
    res, src_key = torch.stack(out), _tagged(xyz_src_BS3HW, "src")
    return _tag(res, "tgt", src_key + _key(G_tgt_src)) if src_key is not None else res


def _same_mask_on_every_plane(obj_mask_BS1HW):
    """The reference hands over S copies of ONE mask (utils/utils.py:323: obj_mask.unsqueeze(1).repeat(B, S, 1, 1, 1)); the fused kernel takes
    the mask once.  An expanded view is recognised by its stride, a materialised repeat by comparing the planes (one pass over the tensor)."""
    if obj_mask_BS1HW.size(1) == 1 or obj_mask_BS1HW.stride(1) == 0:
        return True
    return bool((obj_mask_BS1HW[:, 1:] == obj_mask_BS1HW[:, :1]).all())


def fused_render_applies(mpi_disparity_src, xyz_tgt_BS3HW, xyz_src_BS3HW, G_tgt_src, K_src_inv, K_tgt, use_alpha=False, is_bg_depth_inf=False,
                         hard_flow=False, obj_mask=None):
    """True when render_tgt_rgb_depth may dispatch to the fused kernels: default options, xyz tensors that carry this module's tag for exactly
    these (K_src_inv, disparities, G_tgt_src), and one mask on every plane.  (K_tgt is free: it only enters the homographies.)"""
    if use_alpha or is_bg_depth_inf or hard_flow:
        return False
    src_key, tgt_key = _tagged(xyz_src_BS3HW, "src"), _tagged(xyz_tgt_BS3HW, "tgt")
    if src_key is None or tgt_key is None:
        return False
    want_src = _key(K_src_inv, mpi_disparity_src)
    if src_key != want_src or tgt_key != want_src + _key(G_tgt_src):
        return False
    return obj_mask is None or _same_mask_on_every_plane(obj_mask)


def _render_tgt_fused(mpi_rgb_src, mpi_sigma_src, mpi_disparity_src, G_tgt_src, K_src_inv, K_tgt, obj_mask):
    """The fused form of render_tgt_rgb_depth: Stage B on the two channel-planar tensors in place (mpf_warp_composite_split: homography warp,
    analytic xyz_tgt, front-to-back composite, depth, validity count, rendered mask) + the source-frame flow of the sigma tensor
    (mpf_src_flow) - two launches per batch item instead of the reference-shaped chain (8-channel concatenation, generic sampling of every
    channel, S-deep intermediates)."""
    B, S, _, H, W = mpi_rgb_src.size()
    dev = mpi_rgb_src.device
    rgbs, depths, tms, flows, oms = [], [], [], [], []
    for b in range(B):
        d = host_math.plane_depths(mpi_disparity_src[b])
        H_ts, H_st = host_math.homographies(G_tgt_src[b], K_src_inv[b], K_tgt[b], d)
        quads = ops.mask_quads(obj_mask[b, 0, 0].to(dev, torch.float32).contiguous(), False) if obj_mask is not None else None
        v = ops.warp_composite_split(mpi_rgb_src[b].to(torch.float32), mpi_sigma_src[b].to(torch.float32), quads, H_st, K_src_inv[b], G_tgt_src[b], d)
        rgbs.append(v["rgb"]); depths.append(v["depth"].reshape(1, H, W)); tms.append(v["tgt_mask"].reshape(1, H, W))
        flows.append(ops.src_flow(mpi_sigma_src[b].to(torch.float32), K_src_inv[b], d, H_ts.unsqueeze(0), flow_clip=0.0)[0])
        if obj_mask is not None:
            oms.append(v["objmask"].reshape(1, H, W))
    return torch.stack(rgbs), torch.stack(depths), torch.stack(tms), torch.stack(flows), (torch.stack(oms) if oms else None)


def render_tgt_rgb_depth(H_sampler: HomographySample, mpi_rgb_src, mpi_sigma_src, mpi_disparity_src, xyz_tgt_BS3HW,
                         xyz_src_BS3HW, G_tgt_src, K_src_inv, K_tgt, mpi_flow_src=None, use_alpha=False,
                         is_bg_depth_inf=False, hard_flow=False, obj_mask=None, fused=None):
    """reference :259-349 -> (tgt_rgb_syn Bx3xHxW, tgt_depth_syn Bx1xHxW, tgt_mask Bx1xHxW, flowA2B Bx2xHxW,
    tgt_obj_mask_sync Bx1xHxW).  `mpi_flow_src` is accepted and ignored exactly as in the reference (:267).

    fused (not a reference argument): None = dispatch to the fused kernels when fused_render_applies(...) - the standard call of
    utils/utils.py:291-349 -, else the generic kernels that work on the materialised tensors; False = always generic; True = fused, raising
    if it does not apply.  The two forms differ by fp32 rounding only (the generic one interpolates the xyz_tgt channels bilinearly, the fused
    one evaluates the same affine field at the interpolated coordinate): both sit inside the parity bars against the reference's outputs
    (tests/test_dropin_api.py), tgt_mask is identical."""
    if fused is None or fused:
        ok = fused_render_applies(mpi_disparity_src, xyz_tgt_BS3HW, xyz_src_BS3HW, G_tgt_src, K_src_inv, K_tgt, use_alpha, is_bg_depth_inf,
                                  hard_flow, obj_mask)
        if fused and not ok:
            raise ValueError("render_tgt_rgb_depth(fused=True): the fused kernels need default options, one mask on every plane and xyz tensors built by "
                             "this module's get_src_xyz_from_plane_disparity / get_tgt_xyz_from_plane_disparity for the same K, disparities and pose")
        if ok:
            return _render_tgt_fused(mpi_rgb_src, mpi_sigma_src, mpi_disparity_src, G_tgt_src, K_src_inv, K_tgt, obj_mask)
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
