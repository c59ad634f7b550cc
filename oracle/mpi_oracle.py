"""Python face of the CPU oracle (liboracle.so + the host-side small-matrix maths).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module, and only as the checker / the timed CPU baseline.  The product (mpiflow_amd/) never imports it.

Parity is pinned: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, recorded by
tests/golden/make_golden.py from the reference itself (imported from /root/reference, torch 2.10.0 CPU fp32).

The per-pixel arithmetic lives in plain C (oracle_*.c, each function citing the reference lines it restates).  The
handful of 3x3 / 4x4 matrices per call are computed here with torch-CPU using the reference's own *batched*
expressions, because their last-ulp value depends on the BLAS/LAPACK kernel torch dispatches to
(SURVEY.md §7 hard part 1).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = ctypes.c_void_p


def build(force=False):
    """gcc-compile liboracle.so (and oracle/_ref when the reference tree is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle_math.c", "oracle_mpi.c", "oracle_fwarp.c", "oracle_inpaint.c", "oracle.h", "Makefile")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    ref_src = "/root/reference/external/forward_warping/warping.c"
    if os.path.exists(ref_src) and not os.path.exists(os.path.join(_HERE, "_ref", "libwarping.so")):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_expf.restype = ctypes.c_float
        _LIB.orc_expf.argtypes = [ctypes.c_float]
        _LIB.orc_get_exp_mode.restype = ctypes.c_int
    return _LIB


def ref_warping_lib():
    """The reference's own warping.c compiled by `make -C oracle ref` (None when it was never built)."""
    p = os.path.join(_HERE, "_ref", "libwarping.so")
    return ctypes.CDLL(p) if os.path.exists(p) else None


def set_exp_mode(mode):
    """0 = (float)exp((double)x), closest to the reference's MKL exp; 1 = the HIP kernels' fp32 scheme."""
    lib().orc_set_exp_mode(ctypes.c_int(mode))


def _c(a, dtype=np.float32):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------------------------------------------
# host-side small matrices (torch CPU, reference's batched expressions)
# --------------------------------------------------------------------------------------------------------------

def plane_depths(disparity_S):
    """mpi_depth_src = torch.reciprocal(disparity)   (utils/mpi/mpi_rendering.py:225, :284)"""
    return torch.reciprocal(torch.as_tensor(disparity_S, dtype=torch.float32).reshape(-1)).numpy()


def k_inverse(K):
    """torch.inverse(k_src.double().cpu()).to(fp32)   (utils/utils.py:186-187); K: [3,3] or [1,3,3] -> [3,3]"""
    K = torch.as_tensor(K, dtype=torch.float32).reshape(1, 3, 3)
    return torch.inverse(K.to(torch.float64)).to(torch.float32)[0].numpy()


def homographies(G, K_inv, K_tgt, depth_S):
    """H_tgt_src [S,3,3] and H_src_tgt [S,3,3] exactly as HomographySample.sample builds them
    (utils/mpi/homography_sampler.py:105-122): batched matmuls on the whole [S,3,3] stack, then one batched fp64
    inverse cast back to fp32."""
    S = len(depth_S)
    G = torch.as_tensor(G, dtype=torch.float32).reshape(4, 4)
    K_inv = torch.as_tensor(K_inv, dtype=torch.float32).reshape(1, 3, 3).repeat(S, 1, 1)
    K_tgt = torch.as_tensor(K_tgt, dtype=torch.float32).reshape(1, 3, 3).repeat(S, 1, 1)
    d = torch.as_tensor(depth_S, dtype=torch.float32).reshape(S)
    G_S = G.unsqueeze(0).repeat(S, 1, 1)
    R = G_S[:, 0:3, 0:3]
    t = G_S[:, 0:3, 3]
    n = torch.tensor([0, 0, 1], dtype=torch.float32).unsqueeze(0).repeat(S, 1)
    d33 = d.reshape(S, 1, 1).repeat(1, 3, 3)
    R_tnd = R - torch.matmul(t.unsqueeze(2), n.unsqueeze(1)) / -d33
    H_ts = torch.matmul(K_tgt, torch.matmul(R_tnd, K_inv))
    H_st = torch.inverse(H_ts.to(torch.float64)).to(torch.float32)
    if torch.isnan(H_st).any():
        raise Exception("Matrix inverse contains nan!")
    return H_ts.numpy(), H_st.numpy()


def rot_from_axisangle(vec):
    """geometry.py:114-153 (Rodrigues with angle + 1e-7), vec [B,1,3] -> [B,4,4]"""
    vec = torch.as_tensor(vec, dtype=torch.float32)
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    rot = torch.zeros((vec.shape[0], 4, 4))
    rot[:, 0, 0] = torch.squeeze(x * xC + ca)
    rot[:, 0, 1] = torch.squeeze(xyC - zs)
    rot[:, 0, 2] = torch.squeeze(zxC + ys)
    rot[:, 1, 0] = torch.squeeze(xyC + zs)
    rot[:, 1, 1] = torch.squeeze(y * yC + ca)
    rot[:, 1, 2] = torch.squeeze(yzC - xs)
    rot[:, 2, 0] = torch.squeeze(zxC - ys)
    rot[:, 2, 1] = torch.squeeze(yzC + xs)
    rot[:, 2, 2] = torch.squeeze(z * zC + ca)
    rot[:, 3, 3] = 1
    return rot


def transformation_from_parameters(axisangle, translation, invert=False):
    """geometry.py:79-111: M = T(t) . R(axisangle)  (R^T . T(-t) when invert)"""
    R = rot_from_axisangle(axisangle)
    t = torch.as_tensor(translation, dtype=torch.float32).clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = torch.zeros(t.shape[0], 4, 4)
    T[:, 0, 0] = T[:, 1, 1] = T[:, 2, 2] = T[:, 3, 3] = 1
    T[:, :3, 3, None] = t.contiguous().view(-1, 3, 1)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


def random_pose(rng, ext_cz, base_motions=(0.1, 0.1, 0.1)):
    """generate_random_pose, utils/utils.py:121-156.  `rng` is a random.Random-like object; the draw order
    (3x randrange, 3x random, 3x randrange, 3x random) is part of the contract."""
    import math
    scx = (-1) ** rng.randrange(2)
    scy = (-1) ** rng.randrange(2)
    scz = (-1) ** rng.randrange(2)
    if base_motions[0] == 0.1:
        scz = -1
    else:
        scx, scy, scz = scx * 0.5, scy * 0.5, scz * 0.5
    cx = (rng.random() * 0.1 + base_motions[0]) * scx
    cy = (rng.random() * 0.1 + base_motions[1]) * scy
    cz = (rng.random() * ext_cz + base_motions[2]) * scz
    sax = (-1) ** rng.randrange(2)
    say = (-1) ** rng.randrange(2)
    saz = (-1) ** rng.randrange(2)
    ax = (rng.random() * math.pi / 36.0) * sax
    ay = (rng.random() * math.pi / 36.0) * say
    az = (rng.random() * math.pi / 36.0) * saz
    ang = [ax * 0.4, ay * 0.4, az * 0.4]
    axisangle = torch.from_numpy(np.array([[ang]], dtype=np.float32)).float()
    translation = torch.from_numpy(np.array([[[cx, cy, cz]]][0])).float()
    return transformation_from_parameters(axisangle, translation)[0].numpy()


# --------------------------------------------------------------------------------------------------------------
# generic restatements
# --------------------------------------------------------------------------------------------------------------

def homography_flow(hom_S33, H, W):
    hom = _c(hom_S33).reshape(-1, 9)
    S = hom.shape[0]
    out = np.empty((S, H, W, 2), np.float32)
    lib().orc_homography_flow(_p(hom), S, H, W, _p(out))
    return out


def homography_sample(src_SCHW, hom_src_tgt_S33):
    src = _c(src_SCHW)
    S, C, H, W = src.shape
    hom = _c(hom_src_tgt_S33).reshape(S, 9)
    tgt = np.empty_like(src)
    valid = np.empty((S, H, W), np.uint8)
    flow = np.empty((S, H, W, 2), np.float32)
    lib().orc_homography_sample(_p(src), _p(hom), S, C, H, W, _p(tgt), _p(valid), _p(flow))
    return tgt, valid.astype(bool), flow


def src_xyz(K_inv, depth_S, H, W):
    k = _c(K_inv).reshape(9)
    d = _c(depth_S).reshape(-1)
    out = np.empty((len(d), 3, H, W), np.float32)
    lib().orc_src_xyz(_p(k), _p(d), len(d), H, W, _p(out))
    return out


def transform_xyz(G, xyz_S3HW):
    xyz = _c(xyz_S3HW)
    S = xyz.shape[0]
    N = int(np.prod(xyz.shape[2:]))
    g = _c(G).reshape(16)
    out = np.empty_like(xyz)
    lib().orc_transform_xyz(_p(g), _p(xyz), S, ctypes.c_int64(N), _p(out))
    return out


def volume_render(rgb_S3HW, sigma_S1HW, xyz_S3HW, extra_SEHW=None):
    """-> dict(rgb [3,H,W], depth [H,W], tacc [S,H,W], weights [S,H,W], extra [E,H,W])"""
    sigma = _c(sigma_S1HW)
    xyz = _c(xyz_S3HW)
    S, _, H, W = xyz.shape
    N = H * W
    rgb = _c(rgb_S3HW) if rgb_S3HW is not None else None
    extra = _c(extra_SEHW) if extra_SEHW is not None else None
    E = 0 if extra is None else extra.shape[1]
    out = dict(rgb=np.empty((3, H, W), np.float32) if rgb is not None else None,
               depth=np.empty((H, W), np.float32), tacc=np.empty((S, H, W), np.float32),
               weights=np.empty((S, H, W), np.float32),
               extra=np.empty((E, H, W), np.float32) if E else None)
    lib().orc_volume_render(_p(rgb), _p(sigma), _p(xyz), S, ctypes.c_int64(N), _p(out["rgb"]), _p(out["depth"]),
                            _p(out["tacc"]), _p(out["weights"]), _p(extra), E, _p(out["extra"]))
    return out


def alpha_composition(alpha_S1HW, value_SCHW=None):
    """-> dict(out [C,H,W] | None, weights [S,H,W], cumprod_eps [S,H,W])   (mpi_rendering.py:42-59, :36)"""
    alpha = _c(alpha_S1HW)
    S, _, H, W = alpha.shape
    N = H * W
    val = _c(value_SCHW) if value_SCHW is not None else None
    C = 0 if val is None else val.shape[1]
    out = dict(out=np.empty((C, H, W), np.float32) if C else None, weights=np.empty((S, H, W), np.float32),
               cumprod_eps=np.empty((S, H, W), np.float32))
    lib().orc_alpha_composition(_p(alpha), _p(val), S, C, ctypes.c_int64(N), _p(out["out"]), _p(out["weights"]), _p(out["cumprod_eps"]))
    return out


# --------------------------------------------------------------------------------------------------------------
# fused stages
# --------------------------------------------------------------------------------------------------------------

def src_blend_flow(mpi_S4HW, img_3HW, K_inv, depth_S, hom_tgt_src_PS33, flow_clip=200.0,
                   want_rgba=True, want_planar=False, want_tacc=False):
    mpi = _c(mpi_S4HW)
    S, _, H, W = mpi.shape
    img = _c(img_3HW).reshape(3, H, W)
    hom = np.zeros((0, S, 9), np.float32) if hom_tgt_src_PS33 is None else _c(hom_tgt_src_PS33).reshape(-1, S, 9)
    P = hom.shape[0]
    k = _c(K_inv).reshape(9)
    d = _c(depth_S).reshape(S)
    rgba = np.empty((S, H, W, 4), np.float32) if want_rgba else None
    planar = np.empty((S, 3, H, W), np.float32) if want_planar else None
    tacc = np.empty((S, H, W), np.float32) if want_tacc else None
    flows = np.empty((max(P, 1), 2, H, W), np.float32)[:P]
    lib().orc_src_blend_flow(_p(mpi), _p(img), _p(k), _p(d), _p(hom), P, S, H, W, ctypes.c_float(flow_clip),
                             _p(rgba), _p(planar), _p(tacc), _p(flows))
    return dict(rgba=rgba, rgb_planar=planar, tacc=tacc, flows=flows)


def warp_composite(rgba, obj_mask_HW, hom_src_tgt_S33, K_inv, G, depth_S, interleaved=True, exact_xyz=False):
    a = _c(rgba)
    if interleaved:
        S, H, W, _ = a.shape
    else:
        S, _, H, W = a.shape
    om = _c(obj_mask_HW).reshape(H, W) if obj_mask_HW is not None else None
    hom = _c(hom_src_tgt_S33).reshape(S, 9)
    k = _c(K_inv).reshape(9)
    g = _c(G).reshape(16)
    d = _c(depth_S).reshape(S)
    out = dict(rgb=np.empty((3, H, W), np.float32), depth=np.empty((H, W), np.float32),
               objmask=np.empty((H, W), np.float32) if om is not None else None,
               tgt_mask=np.empty((H, W), np.float32))
    lib().orc_warp_composite(_p(a), int(bool(interleaved)), _p(om), _p(hom), _p(k), _p(g), _p(d), S, H, W,
                             int(bool(exact_xyz)), _p(out["rgb"]), _p(out["depth"]), _p(out["objmask"]),
                             _p(out["tgt_mask"]))
    return out


def to_u8_bgr(img_3HW):
    img = _c(img_3HW)
    _, H, W = img.shape
    out = np.empty((H, W, 3), np.uint8)
    lib().orc_to_u8_bgr(_p(img), H, W, _p(out))
    return out


def merge(frame, frame_dyn, mask, mask_dyn, flow, flow_dyn, obj_mask, thresh=0.99):
    frame, frame_dyn = _c(frame), _c(frame_dyn)
    _, H, W = frame.shape
    flow_mix = np.empty((H, W, 2), np.float32)
    frame_mix = np.empty((H, W, 3), np.uint8)
    fill = np.empty((H, W), np.uint8)
    lib().orc_merge(_p(frame), _p(frame_dyn), _p(_c(mask).reshape(H, W)), _p(_c(mask_dyn).reshape(H, W)),
                    _p(_c(flow).reshape(2, H, W)), _p(_c(flow_dyn).reshape(2, H, W)),
                    _p(_c(obj_mask).reshape(H, W)), ctypes.c_float(np.float32(thresh)), H, W,
                    _p(flow_mix), _p(frame_mix), _p(fill))
    return flow_mix, frame_mix, fill


def merge_depth_ordered(frame, frame_dyn, mask, mask_dyn, depth, depth_dyn, thresh=0.99):
    """The depth-ordered frame of the reference's older module, line by line ("utils/utils copy.py":240-253 quantisation, :278-282 merge,
    :295 mix_mask, :301-303 pick).  numpy, as the reference.  -> (frame_mix_depth [H,W,3] u8 BGR, depth_mask [H,W] bool)"""
    def u8_bgr(f):
        f = np.transpose(np.asarray(f, np.float32), (1, 2, 0))
        return np.clip(np.round(f * 255), a_min=0, a_max=255).astype(np.uint8)[:, :, [2, 1, 0]]
    th = thresh
    mask, mask_dyn = np.asarray(mask, np.float32), np.asarray(mask_dyn, np.float32)
    frame_np, frame_dync_np = u8_bgr(frame), u8_bgr(frame_dyn)
    frame_np[mask < th] = 255
    frame_dync_np[mask_dyn < th] = 255
    frame_mix = frame_dync_np.copy()
    frame_mix[mask >= th] = frame_np[mask >= th]
    mix_mask = np.logical_and(mask, mask_dyn).astype(np.uint8)
    depth_mask = np.logical_and(np.asarray(depth) > np.asarray(depth_dyn), mix_mask)
    frame_mix_depth = frame_mix.copy()
    frame_mix_depth[depth_mask] = frame_dync_np[depth_mask]
    return frame_mix_depth, depth_mask


def render_pair(image_3HW, obj_mask_HW, mpi_S4HW, disparity_S, K, G_cam, G_dyn, thresh=0.99, exact_xyz=False):
    """The whole of render_3dphoto_dynamic (utils/utils.py:159-288) up to the inputs of cv2.inpaint, for given
    poses: cam pose G_cam renders with obj_mask, dynamic pose G_dyn with 1 - obj_mask (sic)."""
    mpi = _c(mpi_S4HW)
    S, _, H, W = mpi.shape
    om = _c(obj_mask_HW).reshape(H, W)
    d = plane_depths(disparity_S)
    k_inv = k_inverse(K)
    Hts_c, Hst_c = homographies(G_cam, k_inv, K, d)
    Hts_d, Hst_d = homographies(G_dyn, k_inv, K, d)
    a = src_blend_flow(mpi, image_3HW, k_inv, d, np.stack([Hts_c, Hts_d]))
    v1 = warp_composite(a["rgba"], om, Hst_c, k_inv, G_cam, d, exact_xyz=exact_xyz)
    v2 = warp_composite(a["rgba"], (1.0 - torch.from_numpy(om)).numpy(), Hst_d, k_inv, G_dyn, d, exact_xyz=exact_xyz)
    flow_mix, frame_mix, fill = merge(v1["rgb"], v2["rgb"], v1["objmask"], v2["objmask"],
                                      a["flows"][0], a["flows"][1], om, thresh)
    return dict(flow_mix=flow_mix, frame_mix=frame_mix, fill_mask=fill, src_np=to_u8_bgr(image_3HW),
                view_cam=v1, view_dyn=v2, flows=a["flows"], rgba=a["rgba"])


# --------------------------------------------------------------------------------------------------------------
# depth -> flow, forward warp (geometry.py, moving_obj.py, warping.c)
# --------------------------------------------------------------------------------------------------------------

def backproject_project(depth_HW, inv_K33, P34):
    depth = _c(depth_HW)
    H, W = depth.shape
    pix = np.empty((H, W, 2), np.float32)
    z = np.empty((H, W), np.float32)
    lib().orc_backproject_project(_p(depth), _p(_c(inv_K33).reshape(9)), _p(_c(P34).reshape(12)), H, W, _p(pix), _p(z))
    return pix, z


def forward_warping(src_u8, idx_i64, idy_i64, z_f32, h, w, use_reference_build=False):
    src = _c(src_u8, np.uint8).reshape(-1)
    idx = _c(idx_i64, np.int64).reshape(-1)
    idy = _c(idy_i64, np.int64).reshape(-1)
    z = _c(z_f32).reshape(-1)
    warped = np.zeros(h * w * 5, np.uint8)
    if use_reference_build:
        ref = ref_warping_lib()
        if ref is None:
            raise RuntimeError("oracle/_ref/libwarping.so not built")
        ref.forward_warping(_p(src), _p(idx), _p(idy), _p(z), _p(warped), ctypes.c_int(h), ctypes.c_int(w))
    else:
        lib().orc_forward_warping(_p(src), _p(idx), _p(idy), _p(z), _p(warped), ctypes.c_int(h), ctypes.c_int(w))
    return warped.reshape(h, w, 5)


def select_truncate(p_static, z_static, p_obj, z_obj, inst_HW):
    inst = _c(inst_HW)
    H, W = inst.shape
    p1 = np.empty((H, W, 2), np.float32)
    z1 = np.empty((H, W), np.float32)
    sx = np.empty((H, W), np.int64)
    sy = np.empty((H, W), np.int64)
    fl = np.empty((H, W, 2), np.float32)
    lib().orc_select_truncate(_p(_c(p_static)), _p(_c(z_static)), _p(_c(p_obj)), _p(_c(z_obj)), _p(inst), H, W,
                              _p(p1), _p(z1), _p(sx), _p(sy), _p(fl))
    return p1, z1, sx, sy, fl


def warp_masks(warped_HW5):
    w5 = _c(warped_HW5, np.uint8)
    H, W, _ = w5.shape
    outs = [np.empty((H, W), np.uint8) for _ in range(5)]
    lib().orc_warp_masks(_p(w5), H, W, *[_p(o) for o in outs])
    return dict(zip(["H", "M", "M'", "P", "H'"], outs))


def moving_object(disp_HW, rgb_HW3_u8, K33, inv_K33, inst_HW, T_obj_44):
    """moveing_object_with_mask (moving_obj.py:16-153) for a given object pose, up to the inputs of cv2.inpaint."""
    disp = torch.as_tensor(_c(disp_HW))
    depth = 1.0 / (disp + 0.005)
    depth[depth > 100] = 100
    H, W = depth.shape
    K4 = torch.zeros(1, 4, 4); K4[0, 3, 3] = 1.0; K4[:, :3, :3] = torch.as_tensor(_c(K33)).reshape(3, 3)
    iK4 = torch.zeros(1, 4, 4); iK4[0, 3, 3] = 1.0; iK4[:, :3, :3] = torch.as_tensor(_c(inv_K33)).reshape(3, 3)
    T1 = transformation_from_parameters(torch.zeros(1, 1, 3), torch.zeros(1, 3))
    P1 = torch.matmul(K4, T1)[:, :3, :][0].numpy()
    Pi = torch.matmul(K4, torch.as_tensor(_c(T_obj_44)).reshape(1, 4, 4))[:, :3, :][0].numpy()
    ps, zs = backproject_project(depth.numpy(), iK4[0, :3, :3].numpy(), P1)
    po, zo = backproject_project(depth.numpy(), iK4[0, :3, :3].numpy(), Pi)
    p1, z1, sx, sy, fl = select_truncate(ps, zs, po, zo, inst_HW)
    warped = forward_warping(_c(rgb_HW3_u8, np.uint8), sx, sy, z1, H, W)
    return dict(p1=p1, z1=z1, safe_x=sx, safe_y=sy, flow01=fl, warped=warped, masks=warp_masks(warped))


# --------------------------------------------------------------------------------------------------------------
# cv2.inpaint / cv2.dilate restated (oracle_inpaint.c) - PARITY UNPINNED: third-party OpenCV, not installed here
# --------------------------------------------------------------------------------------------------------------

INPAINT_NS, INPAINT_TELEA = 0, 1


def inpaint(img_u8, mask_u8, radius, method, reading=0):
    """cv2.inpaint(img, mask, radius, method) for u8 [H,W,3] / [H,W,1] / [H,W] images (utils/utils.py:284-286, moving_obj.py:162).
    reading: 0 = OpenCV's unqualified sqrt / fabs on floats as the float overloads (default), 1 = as the double functions - the one
    open question of this restatement, see oracle_inpaint.c and tests/golden/inpaint_reading_exhibit.npz"""
    lib().orc_inpaint_set_reading(int(reading))
    try:
        return _inpaint(img_u8, mask_u8, radius, method)
    finally:
        lib().orc_inpaint_set_reading(0)


def _inpaint(img_u8, mask_u8, radius, method):
    img = _c(img_u8, np.uint8)
    H, W = img.shape[:2]
    C = 1 if img.ndim == 2 else img.shape[2]
    mask = _c(mask_u8, np.uint8).reshape(H, W)
    out = np.empty_like(img)
    rc = lib().orc_inpaint(_p(img), _p(mask), H, W, C, ctypes.c_double(radius), int(method), _p(out))
    if rc != 0:
        raise MemoryError("orc_inpaint")
    return out


def dilate3x3(img_u8):
    img = _c(img_u8, np.uint8)
    H, W = img.shape
    out = np.empty_like(img)
    lib().orc_dilate3x3(_p(img), H, W, _p(out))
    return out


# --------------------------------------------------------------------------------------------------------------
# input stage (utils/utils.py:35-52, gen_3dphoto_dynamic_v2.py:82-89, :101-105)
# --------------------------------------------------------------------------------------------------------------

def resize_bilinear_ac(x_CHW, H, W):
    """F.interpolate(x[None], size=(H,W), mode='bilinear', align_corners=True)[0] for fp32 [C,h,w]"""
    x = _c(x_CHW)
    C, h, w = x.shape
    out = np.empty((C, H, W), np.float32)
    lib().orc_resize_bilinear_ac(_p(x), C, h, w, H, W, _p(out))
    return out


def prepare_inputs(rgb_u8_hw3=None, disp_u8_hw=None, ids_u8_hw=None, obj_index=0, size=None):
    """u8 arrays as decoded from the files -> the resized float tensors of gen_3dphoto_dynamic_v2.py:82-89 / :101-105"""
    H, W = size
    out = dict(image=None, disp=None, mask=None)
    if rgb_u8_hw3 is not None:
        img = np.ascontiguousarray(np.asarray(rgb_u8_hw3, np.uint8).transpose(2, 0, 1)).astype(np.float32) / np.float32(255)   # ToTensor
        out["image"] = resize_bilinear_ac(img, H, W)
    if disp_u8_hw is not None:
        d = (np.asarray(disp_u8_hw, np.uint8) / 255).astype(np.float32)                    # float64 division, then .float()
        out["disp"] = resize_bilinear_ac(d[None], H, W)[0]
    if ids_u8_hw is not None:
        m = (np.asarray(ids_u8_hw) == obj_index).astype(np.float32)
        out["mask"] = resize_bilinear_ac(m[None], H, W)[0]
    return out
