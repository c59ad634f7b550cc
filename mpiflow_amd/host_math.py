"""Host-side small-matrix maths of the MPI-Flow path (a few 3x3 / 4x4 matrices per image pair).

These stay on the host, in torch-CPU, using the reference's *batched* expressions on purpose: the composite
amplifies a 1-ulp change of a homography entry into >1e-4 on the rendered RGB (SURVEY.md §7 hard part 1), and the
last ulp of `[S,3,3] @ [S,3,3]` / batched fp64 `inverse` depends on the BLAS/LAPACK kernel torch dispatches to.
Everything per-pixel happens in the HIP kernels; what is computed here is packed into the `d_params` buffer of
include/mpiflow_hip.h (S x 16 floats).
"""
import math

import numpy as np
import torch

PARAMS_HEADER = 32
PLANE_RECORD = 16


def _cpu32(x):
    if isinstance(x, torch.Tensor):
        return x.detach().to(device="cpu", dtype=torch.float32)
    return torch.as_tensor(np.asarray(x), dtype=torch.float32)


def plane_depths(disparity):
    """mpi_depth_src = torch.reciprocal(disparity)   (reference utils/mpi/mpi_rendering.py:225, :284) -> [S]"""
    return torch.reciprocal(_cpu32(disparity).reshape(-1))


def inverse(matrices):
    """Batched inverse with the reference's retry-on-NaN behaviour (utils/mpi/homography_sampler.py:6-27):
    up to 5 attempts, then Exception("Matrix inverse contains nan!").  Runs on CPU like the reference's."""
    inv = None
    tries = 5
    m_cpu = matrices.detach().cpu()
    while inv is None or torch.isnan(inv).any():
        inv = torch.inverse(m_cpu)
        tries -= 1
        if tries == 0:
            break
    if torch.isnan(inv).any():
        raise Exception("Matrix inverse contains nan!")
    return inv.to(matrices.device)


def k_inverse(K):
    """torch.inverse(k_src.double().cpu()).to(dtype)   (utils/utils.py:186-187).  K [..,3,3] -> [3,3] fp32 CPU"""
    return torch.inverse(_cpu32(K).reshape(3, 3).to(torch.float64)).to(torch.float32)


def homographies(G_tgt_src, K_src_inv, K_tgt, depth_S):
    """(H_tgt_src [S,3,3], H_src_tgt [S,3,3]) fp32 CPU, built as HomographySample.sample does
    (utils/mpi/homography_sampler.py:105-122): H_tgt_src = K_tgt (R - t n^T / -d) K_src^-1 with batched matmuls on the
    [S,3,3] stack, H_src_tgt = one batched fp64 inverse cast back to fp32."""
    d = _cpu32(depth_S).reshape(-1)
    S = d.numel()
    G = _cpu32(G_tgt_src).reshape(4, 4).unsqueeze(0).repeat(S, 1, 1)
    Kinv = _cpu32(K_src_inv).reshape(1, 3, 3).repeat(S, 1, 1)
    Kt = _cpu32(K_tgt).reshape(1, 3, 3).repeat(S, 1, 1)
    R = G[:, 0:3, 0:3]
    t = G[:, 0:3, 3]
    n = torch.tensor([0, 0, 1], dtype=torch.float32).unsqueeze(0).repeat(S, 1)
    d33 = d.reshape(S, 1, 1).repeat(1, 3, 3)
    R_tnd = R - torch.matmul(t.unsqueeze(2), n.unsqueeze(1)) / -d33
    H_ts = torch.matmul(Kt, torch.matmul(R_tnd, Kinv))
    H_st = inverse(H_ts.to(torch.float64)).to(torch.float32)
    return H_ts, H_st


def homographies_multi(G_list, K_src_inv, K_tgt, depth_S):
    """homographies() for P poses in ONE batched evaluation ([P*S,3,3] stacks: every matrix goes through the same per-matrix
    matmul / LU code as in a [S,3,3] batch, so the results are bit-identical - asserted against the goldens).
    -> (H_tgt_src [P,S,3,3], H_src_tgt [P,S,3,3])"""
    d = _cpu32(depth_S).reshape(-1)
    S, P = d.numel(), len(G_list)
    G = torch.stack([_cpu32(g).reshape(4, 4) for g in G_list]).unsqueeze(1).repeat(1, S, 1, 1).reshape(P * S, 4, 4)
    Kinv = _cpu32(K_src_inv).reshape(1, 3, 3).repeat(P * S, 1, 1)
    Kt = _cpu32(K_tgt).reshape(1, 3, 3).repeat(P * S, 1, 1)
    R = G[:, 0:3, 0:3]
    t = G[:, 0:3, 3]
    n = torch.tensor([0, 0, 1], dtype=torch.float32).unsqueeze(0).repeat(P * S, 1)
    d33 = d.repeat(P).reshape(P * S, 1, 1).repeat(1, 3, 3)
    R_tnd = R - torch.matmul(t.unsqueeze(2), n.unsqueeze(1)) / -d33
    H_ts = torch.matmul(Kt, torch.matmul(R_tnd, Kinv))
    H_st = inverse(H_ts.to(torch.float64)).to(torch.float32)
    return H_ts.reshape(P, S, 3, 3), H_st.reshape(P, S, 3, 3)


def pack_params(K_inv=None, G=None, homs=None, depths=None, records=None):
    """Build the `d_params` host image (float32 CPU tensor) of include/mpiflow_hip.h.
    homs: [R,3,3] (record r = s*P + p), depths: [R] (already repeated per record)."""
    if records is None:
        records = 0 if homs is None else int(homs.shape[0])
        if homs is None and depths is not None:
            records = int(depths.numel())
    buf = torch.zeros(PARAMS_HEADER + PLANE_RECORD * max(records, 1), dtype=torch.float32)
    if K_inv is not None:
        buf[0:9] = _cpu32(K_inv).reshape(9)
    if G is not None:
        buf[9:21] = _cpu32(G).reshape(4, 4)[0:3, :].reshape(12)
    if records:
        rec = buf[PARAMS_HEADER:].view(-1, PLANE_RECORD)
        if homs is not None:
            rec[:records, 0:9] = _cpu32(homs).reshape(records, 9)
        if depths is not None:
            rec[:records, 9] = _cpu32(depths).reshape(records)
    return buf


# ---- poses (geometry.py:79-153, utils/utils.py:121-156) -----------------------------------------------------------

def rot_from_axisangle(vec):
    """Axis-angle [B,1,3] -> rotation [B,4,4]: Rodrigues with the reference's regularised axis (angle + 1e-7) and its exact
    operand pairing (geometry.py:114-153): diagonal u_i*(u_i*C) + cos, off-diagonals u_i*(u_j*C) -/+ u_k*sin with the
    products taken as x*(y*C), y*(z*C), z*(x*C) - the pairing decides the last ulp."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    cos_a, sin_a = torch.cos(angle), torch.sin(angle)
    C = 1 - cos_a
    u = [axis[..., i].unsqueeze(1) for i in range(3)]
    u_sin = [c * sin_a for c in u]
    u_C = [c * C for c in u]
    rot = torch.zeros((vec.shape[0], 4, 4)).to(device=vec.device)
    for i in range(3):
        rot[:, i, i] = torch.squeeze(u[i] * u_C[i] + cos_a)
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):          # (i, j) off-diagonal pair, k the remaining axis
        prod = u[i] * u_C[j]
        rot[:, i, j] = torch.squeeze(prod - u_sin[k])
        rot[:, j, i] = torch.squeeze(prod + u_sin[k])
    rot[:, 3, 3] = 1
    return rot


def get_translation_matrix(translation_vector):
    """Translation [B,3] (or [B,1,3]) -> [B,4,4] (geometry.py:98-111)."""
    T = torch.zeros(translation_vector.shape[0], 4, 4).to(device=translation_vector.device)
    t = translation_vector.contiguous().view(-1, 3, 1)
    T[:, 0, 0] = 1
    T[:, 1, 1] = 1
    T[:, 2, 2] = 1
    T[:, 3, 3] = 1
    T[:, :3, 3, None] = t
    return T


def transformation_from_parameters(axisangle, translation, invert=False):
    """(axisangle [B,1,3], translation [B,3]) -> 4x4: M = T.R, or R^T.T(-t) when invert (geometry.py:79-95)."""
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t *= -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


# The reference carries three copies of its pose sampler that differ only in constants:
#   "v2"    utils/utils.py:121-156         (gen_3dphoto_dynamic_v2.py; the default)
#   "coco"  utils/utils_coco.py:121-154
#   "copy"  "utils/utils copy.py":121-160
# forward_base: the base_motions[0] value that marks the CAMERA draw (z forced forward); any other draw is an object motion
#               and gets its translation signs halved - "coco" has neither rule.  half_angle_signs_unless: base_motions[0]
#               value that keeps full-size rotation signs (None: never halved).
POSE_PROFILES = {
    "v2":   dict(default_base=(0.1, 0.1, 0.1), forward_base=0.1, cz_from_ext=True, xy_scale=None, half_angle_signs_unless=None, angle_scale=0.4),
    "coco": dict(default_base=(0.1, 0.1, 0.1), forward_base=None, cz_from_ext=False, xy_scale=None, half_angle_signs_unless=0.05, angle_scale=0.5),
    "copy": dict(default_base=(0.05, 0.05, 0.05), forward_base=0.05, cz_from_ext=False, xy_scale=0.3, half_angle_signs_unless=0.05, angle_scale=0.2),
}


def draw_pose_parameters(ext_cz=0.1, base_motions=None, rng=None, profile="v2"):
    """The 12 draws of one pose and the reference's scalar arithmetic on them -> (axis-angle [3], translation [3]) as Python
    floats.  Draw order - 3x randrange(2), 3x random(), 3x randrange(2), 3x random() on Python's `random` - is part of the
    contract (seeded runs reproduce the reference's poses), and so is the order of the float operations."""
    import random as _random
    rng = rng or _random
    prof = POSE_PROFILES[profile]
    base = prof["default_base"] if base_motions is None else base_motions
    sign_t = [(-1) ** rng.randrange(2) for _ in range(3)]
    if prof["forward_base"] is not None:
        if base[0] == prof["forward_base"]:
            sign_t[2] = -1                                   # "most cameras move forward in kitti"
        else:
            sign_t = [v * 0.5 for v in sign_t]                # object motion
    width = [0.1, 0.1, ext_cz if prof["cz_from_ext"] else 0.1]
    t = [(rng.random() * width[i] + base[i]) * sign_t[i] for i in range(3)]
    if prof["xy_scale"] is not None:
        t = [t[0] * prof["xy_scale"], t[1] * prof["xy_scale"], t[2]]
    sign_a = [(-1) ** rng.randrange(2) for _ in range(3)]
    if prof["half_angle_signs_unless"] is not None and not base[0] == prof["half_angle_signs_unless"]:
        sign_a = [v * 0.5 for v in sign_a]
    ang = [(rng.random() * math.pi / 36.0) * sign_a[i] for i in range(3)]
    ang = [v * prof["angle_scale"] for v in ang]
    return ang, t


def poses_from_parameters(params):
    """[(axis-angle, translation), ...] -> [n,4,4] fp32 CPU, one batched transformation_from_parameters (bit-identical to n
    single evaluations: element-wise ops and per-matrix 4x4 products; asserted against the reference's pose goldens)."""
    aa = torch.from_numpy(np.array([p[0] for p in params], dtype=np.float32).reshape(-1, 1, 3)).float()
    tr = torch.from_numpy(np.array([p[1] for p in params]).reshape(-1, 1, 3)).float()
    return transformation_from_parameters(aa, tr)


def generate_random_pose(ext_cz=0.1, base_motions=None, rng=None, profile="v2"):
    """Random camera extrinsic (utils/utils.py:121-156 and its two variants) -> [4,4] fp32 CPU tensor.
    `rng`: a random.Random; default = the module-level generator the reference uses."""
    ang, t = draw_pose_parameters(ext_cz, base_motions, rng, profile)
    axisangle = torch.from_numpy(np.array([[ang]], dtype=np.float32)).float()
    translation = torch.from_numpy(np.array([[t]])).float()
    return transformation_from_parameters(axisangle, translation)[0]
