#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; imports it read-only through ref_harness.py with the
stubs listed there).  The reference's Python never travels; only the .npz data written here is committed.

    python tests/golden/make_golden.py            # everything (c2 takes ~5 min and ~8 GB)
    python tests/golden/make_golden.py tiny odd   # a subset

Every file records torch/numpy versions (the reference's fp32 arithmetic is executed by torch-CPU ATen/MKL kernels,
so the goldens are "the reference on torch 2.10.0 CPU, this image").  Inputs come from mpiflow_amd.synth
(seeded numpy, reproducible anywhere); for the big shapes only the seed is stored, not the tensors.

Margin check (SURVEY §7 hard part 2): thresholded masks are only asserted bit-exact away from the threshold; each
file stores `margin_px_*`, the pixels whose reference value lies within 1e-5 of 0.99, so tests can report them
instead of hiding them.
"""
import hashlib
import os
import random
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_harness  # noqa: E402
from mpiflow_amd import synth  # noqa: E402

R = ref_harness.modules()
THRESH = 0.99
MARGIN = 1e-5


def meta():
    return dict(torch_version=torch.__version__, numpy_version=np.__version__,
                cpu_capability=torch.backends.cpu.get_cpu_capability(),
                generated_unix=int(time.time()))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_pixels(H, W, n=4096, seed=99):
    rs = np.random.RandomState(seed)
    n = min(n, H * W)
    idx = rs.choice(H * W, size=n, replace=False)
    corners = np.array([0, W - 1, (H - 1) * W, H * W - 1], dtype=np.int64)
    border = np.concatenate([np.arange(0, W, max(W // 16, 1)), (H - 1) * W + np.arange(0, W, max(W // 16, 1)),
                             np.arange(0, H, max(H // 16, 1)) * W, np.arange(0, H, max(H // 16, 1)) * W + W - 1])
    return np.unique(np.concatenate([idx, corners, border])).astype(np.int64)


def poses_for(seed, ext_cz=0.15):
    """The two poses render_3dphoto_dynamic would draw (utils/utils.py:207-208) under random.seed(seed)."""
    random.seed(seed)
    G_dyn = R.utils.generate_random_pose(ext_cz)
    G_cam = R.utils.generate_random_pose(ext_cz, base_motions=[0, 0, 0])
    return G_cam, G_dyn


class Opt:
    ext_cz = 0.15


def run_pair(inp, pose_seed):
    """Run the reference's render_3dphoto_dynamic and collect its outputs plus the inputs of cv2.inpaint."""
    S, _, H, W = inp["mpi"].shape
    mpi = torch.from_numpy(inp["mpi"])[None]
    disp = torch.from_numpy(inp["disparity"])[None]
    K = torch.from_numpy(inp["K"])[None]
    img = torch.from_numpy(inp["image"])[None]
    om = torch.from_numpy(inp["obj_mask"])[None, None]
    G_cam, G_dyn = poses_for(pose_seed)
    random.seed(pose_seed)  # render_3dphoto_dynamic redraws the same two poses
    ref_harness.captured.clear()
    t0 = time.time()
    flow_mix, src_np, inpainted, _ = R.utils.render_3dphoto_dynamic(Opt, img, om, None, mpi, disp, K, K, name="golden")
    dt = time.time() - t0
    cap = ref_harness.captured["inpaint"]
    return dict(G_cam=G_cam.numpy(), G_dyn=G_dyn.numpy(), flow_mix=flow_mix, src_np=src_np,
                frame_mix=cap["img"], fill_mask=cap["mask"].astype(np.uint8), seconds=dt)


def run_views(inp, G_cam, G_dyn):
    """The two render_novel_view_dynamic calls (utils/utils.py:210-236) on the blended stack, un-thresholded."""
    S, _, H, W = inp["mpi"].shape
    mpi = torch.from_numpy(inp["mpi"])[None]
    disp = torch.from_numpy(inp["disparity"])[None]
    K = torch.from_numpy(inp["K"])[None]
    img = torch.from_numpy(inp["image"])[None]
    om = torch.from_numpy(inp["obj_mask"])[None, None]
    hs = R.homography_sampler.HomographySample(H, W, torch.device("cpu"))
    k_inv = torch.inverse(K.double()).float()
    rgb, sig = mpi[:, :, 0:3], mpi[:, :, 3:]
    xyz_src = R.mpi_rendering.get_src_xyz_from_plane_disparity(hs.meshgrid, disp, k_inv)
    _, _, bw, _, _, _ = R.mpi_rendering.render(rgb, sig, xyz_src, use_alpha=False, is_bg_depth_inf=False)
    rgb_b = bw * img.unsqueeze(1) + (1 - bw) * rgb
    del xyz_src
    out = dict(k_inv=k_inv[0].numpy(), blend_weights=bw[0, :, 0].numpy(), rgb_blended=rgb_b[0].numpy())
    for tag, m, G in (("cam", om, torch.from_numpy(G_cam)), ("dyn", 1 - om, torch.from_numpy(G_dyn))):
        frame, depth, flow, mask = R.utils.render_novel_view_dynamic(m, rgb_b, sig, disp, G, k_inv, K, K, None, hs)
        out[tag] = dict(rgb=frame[0].numpy(), depth=depth[0, 0].numpy(), flow=flow[0].numpy(), objmask=mask[0, 0].numpy())
    return out


def margin_pixels(a):
    return np.flatnonzero(np.abs(a.astype(np.float64).ravel() - np.float64(np.float32(THRESH))) < MARGIN).astype(np.int64)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    arrays.update({"meta_" + k: np.array(v) for k, v in meta().items()})
    np.savez_compressed(path, **arrays)
    print("wrote %s  (%.1f KB)" % (path, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------------------------------------------

def gen_small(name, S, H, W, kind, seed, pose_seed, full_intermediates):
    """Full-tensor goldens: inputs, poses, every intermediate of one posed view, both views, merge products."""
    inp = synth.make_inputs(S, H, W, seed=seed, kind=kind)
    pair = run_pair(inp, pose_seed)
    views = run_views(inp, pair["G_cam"], pair["G_dyn"])
    arrays = dict(S=S, H=H, W=W, kind=kind, seed=seed, pose_seed=pose_seed,
                  mpi=inp["mpi"], disparity=inp["disparity"], image=inp["image"], obj_mask=inp["obj_mask"], K=inp["K"],
                  G_cam=pair["G_cam"], G_dyn=pair["G_dyn"], k_inv=views["k_inv"],
                  flow_mix=pair["flow_mix"], src_np=pair["src_np"], frame_mix=pair["frame_mix"], fill_mask=pair["fill_mask"])
    for tag in ("cam", "dyn"):
        for k, v in views[tag].items():
            arrays["%s_%s" % (tag, k)] = v
        arrays["margin_px_%s" % tag] = margin_pixels(views[tag]["objmask"])
    arrays["margin_px_obj"] = margin_pixels(inp["obj_mask"])
    arrays["blend_weights"] = views["blend_weights"]
    arrays["rgb_blended"] = views["rgb_blended"]
    if full_intermediates:
        # one posed view, function by function (SURVEY §8(b) signatures)
        mpi = torch.from_numpy(inp["mpi"])[None]
        disp = torch.from_numpy(inp["disparity"])[None]
        K = torch.from_numpy(inp["K"])[None]
        om = torch.from_numpy(inp["obj_mask"])[None, None]
        G = torch.from_numpy(pair["G_cam"])
        hs = R.homography_sampler.HomographySample(H, W, torch.device("cpu"))
        k_inv = torch.inverse(K.double()).float()
        d = torch.reciprocal(disp)
        xyz_src = R.mpi_rendering.get_src_xyz_from_plane_disparity(hs.meshgrid, disp, k_inv)
        xyz_tgt = R.mpi_rendering.get_tgt_xyz_from_plane_disparity(xyz_src, G[None])
        rgb_b = torch.from_numpy(views["rgb_blended"])[None]
        sig = mpi[:, :, 3:]
        cat = torch.cat((rgb_b, sig, xyz_tgt, om.unsqueeze(1).repeat(1, S, 1, 1, 1)), dim=2)[0]
        GS, KiS, KS = G[None].repeat(S, 1, 1), k_inv.repeat(S, 1, 1), K.repeat(S, 1, 1)
        tgt, valid, fB2A = hs.sample(cat, d[0], GS, KiS, KS)
        fA2B = hs.sample_inverse(cat, d[0], GS, KiS, KS)
        # the homographies themselves, rebuilt with the reference's expressions (homography_sampler.py:105-122)
        n = hs.n.unsqueeze(0).repeat(S, 1)
        d33 = d[0].reshape(S, 1, 1).repeat(1, 3, 3)
        R_tnd = GS[:, :3, :3] - torch.matmul(GS[:, :3, 3].unsqueeze(2), n.unsqueeze(1)) / -d33
        H_ts = torch.matmul(KS, torch.matmul(R_tnd, KiS))
        H_st = R.homography_sampler.inverse(H_ts.to(torch.float64)).to(torch.float32)
        r_rgb, r_depth, r_tmask, r_flow, r_om = R.mpi_rendering.render_tgt_rgb_depth(
            hs, rgb_b, sig, disp, xyz_tgt, xyz_src, G[None], k_inv, K, None,
            obj_mask=om.unsqueeze(1).repeat(1, S, 1, 1, 1))
        _, _, _, w_src, _, _ = R.mpi_rendering.render(mpi[:, :, :3], sig, xyz_src)
        arrays.update(depth_S=d[0].numpy(), meshgrid=hs.meshgrid.numpy(), xyz_src=xyz_src[0].numpy(),
                      xyz_tgt_cam=xyz_tgt[0].numpy(), H_tgt_src_cam=H_ts.numpy(), H_src_tgt_cam=H_st.numpy(),
                      sample_tgt=tgt.numpy(), sample_valid=valid.numpy(), sample_flowB2A=fB2A.numpy(),
                      sample_inverse_flow=fA2B.numpy(), weights_src=w_src[0, :, 0].numpy(),
                      rtd_rgb=r_rgb[0].numpy(), rtd_depth=r_depth[0, 0].numpy(), rtd_tgt_mask=r_tmask[0, 0].numpy(),
                      rtd_flow_unclipped=r_flow[0].numpy(), rtd_objmask=r_om[0, 0].numpy())
    save(name, **arrays)


def gen_big(name, S, H, W, kind, seed, pose_seed, stack_px=None, require_empty_margin=False):
    """Config-shape goldens: seeds + poses + values at a fixed pixel sample + packed bit masks.
    stack_px: keep the per-plane [S, ...] samples (blend weights, blended stack) at only the first `stack_px` sample
    pixels (keeps the 128-plane file small)."""
    inp = synth.make_inputs(S, H, W, seed=seed, kind=kind)
    t0 = time.time()
    pair = run_pair(inp, pose_seed)
    print("  reference render_3dphoto_dynamic %dx%dx%d: %.1f s" % (S, H, W, pair["seconds"]))
    views = run_views(inp, pair["G_cam"], pair["G_dyn"])
    px = sample_pixels(H, W)
    th = np.float32(THRESH)
    arrays = dict(S=S, H=H, W=W, kind=kind, seed=seed, pose_seed=pose_seed, K=inp["K"], disparity=inp["disparity"],
                  G_cam=pair["G_cam"], G_dyn=pair["G_dyn"], k_inv=views["k_inv"], sample_px=px,
                  ref_seconds=pair["seconds"], sha_inputs_mpi=sha(inp["mpi"]), sha_inputs_image=sha(inp["image"]),
                  flow_mix_px=pair["flow_mix"].reshape(-1, 2)[px], frame_mix_px=pair["frame_mix"].reshape(-1, 3)[px],
                  src_np_px=pair["src_np"].reshape(-1, 3)[px],
                  fill_mask_bits=np.packbits(pair["fill_mask"].ravel()), fill_mask_count=int(pair["fill_mask"].sum()),
                  sha_frame_mix=sha(pair["frame_mix"]), sha_fill_mask=sha(pair["fill_mask"]), sha_src_np=sha(pair["src_np"]),
                  flow_mix_stats=np.array([pair["flow_mix"].min(), pair["flow_mix"].max(), pair["flow_mix"].astype(np.float64).sum()]),
                  blend_weights_px=views["blend_weights"].reshape(S, -1)[:, px[:stack_px]],
                  rgb_blended_px=views["rgb_blended"].reshape(S, 3, -1)[:, :, px[:stack_px]])
    for tag in ("cam", "dyn"):
        v = views[tag]
        arrays["%s_rgb_px" % tag] = v["rgb"].reshape(3, -1)[:, px]
        arrays["%s_depth_px" % tag] = v["depth"].ravel()[px]
        arrays["%s_flow_px" % tag] = v["flow"].reshape(2, -1)[:, px]
        arrays["%s_objmask_px" % tag] = v["objmask"].ravel()[px]
        arrays["%s_mask_bits" % tag] = np.packbits((v["objmask"] >= th).ravel())
        arrays["margin_px_%s" % tag] = margin_pixels(v["objmask"])
        arrays["%s_stats" % tag] = np.array([v["rgb"].astype(np.float64).sum(), v["flow"].astype(np.float64).sum(),
                                              v["objmask"].astype(np.float64).sum()])
    arrays["margin_px_obj"] = margin_pixels(inp["obj_mask"])
    print("  margin pixels: cam %d dyn %d obj %d ; total %.1f s" % (len(arrays["margin_px_cam"]), len(arrays["margin_px_dyn"]),
                                                                       len(arrays["margin_px_obj"]), time.time() - t0))
    if require_empty_margin:
        # the *_opaque goldens exist to compare thresholded masks with NO exclusion: the reference's own values must keep twice the margin
        for tag in ("cam", "dyn"):
            near = int((np.abs(views[tag]["objmask"].astype(np.float64) - np.float64(np.float32(THRESH))) < 2 * MARGIN).sum())
            assert near == 0, "%s: %d reference mask values of view %s lie within 2e-5 of the threshold - pick another seed (tools: oracle search)" % (name, near, tag)
        assert len(arrays["margin_px_obj"]) == 0
    save(name, **arrays)


def gen_fwarp(name, H, W, seed, full):
    """moveing_object_with_mask (moving_obj.py:16-168) with its C forward warp = the reference's own warping.c."""
    rs = np.random.RandomState(seed)
    # disparity: smooth background + a near rectangle (so background pixels collide behind the moved object)
    base = synth._upsample(rs.rand(max(H // 16, 2), max(W // 16, 2)), H, W) * 0.3 + 0.05
    inst = np.zeros((H, W), np.float32)
    inst[H // 3: 2 * H // 3, W // 3: 2 * W // 3] = 1.0
    disp = (base + 0.5 * inst).astype(np.float32)
    rgb = np.floor(rs.rand(H, W, 3) * 256).astype(np.float32)     # "np float holding 0..255"
    K = synth.intrinsics(H, W)
    inv_K = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    # run the reference, capturing what it hands to the C routine and to cv2.inpaint
    import ctypes
    mo = R.moving_obj
    seen = {}
    real_warp = mo.warp

    def spy(src, idx, idy, z, warped, h, w):
        n = h.value * w.value
        seen["src"] = np.ctypeslib.as_array(ctypes.cast(src, ctypes.POINTER(ctypes.c_uint8)), (n * 3,)).copy()
        seen["idx"] = np.ctypeslib.as_array(ctypes.cast(idx, ctypes.POINTER(ctypes.c_int64)), (n,)).copy()
        seen["idy"] = np.ctypeslib.as_array(ctypes.cast(idy, ctypes.POINTER(ctypes.c_int64)), (n,)).copy()
        seen["z"] = np.ctypeslib.as_array(ctypes.cast(z, ctypes.POINTER(ctypes.c_float)), (n,)).copy()
        real_warp(src, idx, idy, z, warped, h, w)
        seen["warped"] = np.ctypeslib.as_array(ctypes.cast(warped, ctypes.POINTER(ctypes.c_uint8)), (n * 5,)).copy()

    mo.warp = spy
    random.seed(seed)
    os.makedirs("temp", exist_ok=True)
    cwd = os.getcwd()
    os.chdir("/tmp")
    os.makedirs("temp", exist_ok=True)
    try:
        mo.moveing_object_with_mask(None, torch.from_numpy(disp)[None, None], rgb, torch.from_numpy(K),
                                    torch.from_numpy(inv_K), torch.from_numpy(inst)[None, None], 0)
    finally:
        os.chdir(cwd)
        mo.warp = real_warp
    # replay the RNG to recover the object translation (moving_obj.py:81-98; angles are overwritten to 0)
    random.seed(seed)
    t = [random.random() * 0.05 + 0.05, -1 * (random.random() * 0.05 + 0.05), random.random() * 0.05 + 0.05]
    Ti = R.geometry.transformation_from_parameters(torch.zeros(1, 1, 3), torch.from_numpy(np.array([t])).float())[0].numpy()
    inp_mask = ref_harness.captured["inpaint"]["mask"]           # 1 - H
    warped = seen["warped"].reshape(H, W, 5)
    arrays = dict(H=H, W=W, seed=seed, K=K, inv_K=inv_K, T_obj=Ti, obj_translation=np.array(t),
                  sha_warped=sha(warped), sha_idx=sha(seen["idx"]), sha_idy=sha(seen["idy"]), sha_z=sha(seen["z"]),
                  n_valid=int(warped[..., 3].sum()), n_single=int(warped[..., 4].sum()))
    if full:
        arrays.update(disp=disp, rgb=rgb.astype(np.uint8), inst=inst, safe_x=seen["idx"].reshape(H, W),
                      safe_y=seen["idy"].reshape(H, W), z1=seen["z"].reshape(H, W), warped=warped,
                      inpaint_mask=inp_mask.reshape(H, W))
    else:
        px = sample_pixels(H, W)
        arrays.update(sample_px=px, safe_x_px=seen["idx"][px], safe_y_px=seen["idy"][px], z1_px=seen["z"][px],
                      warped_px=warped.reshape(-1, 5)[px], valid_bits=np.packbits(warped[..., 3].ravel()),
                      single_bits=np.packbits(warped[..., 4].ravel()))
    save(name, **arrays)


def gen_collision_stress(name="fwarp_stress", h=40, w=56, seed=5):
    """Heavy-collision known-answer test for warping.c alone: random targets incl. border pile-ups, ties in z,
    and z values equal to the 1000 sentinel."""
    import ctypes
    rs = np.random.RandomState(seed)
    n = h * w
    idx = rs.randint(0, max(w // 4, 1), size=n).astype(np.int64)
    idy = rs.randint(0, max(h // 4, 1), size=n).astype(np.int64)
    idx[: n // 8] = 0
    idy[: n // 16] = 0                                  # pile-ups on a border column / corner
    z = (rs.randint(0, 6, size=n).astype(np.float32)) * 0.5 + 1.0   # many exact ties
    z[rs.rand(n) < 0.02] = 1000.0                       # sentinel collisions
    z[rs.rand(n) < 0.01] = 2000.0
    src = rs.randint(0, 256, size=n * 3).astype(np.uint8)
    warped = np.zeros(n * 5, np.uint8)
    lib = ctypes.CDLL(ref_harness.REF_WARP_SO)
    lib.forward_warping(src.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p),
                        idy.ctypes.data_as(ctypes.c_void_p), z.ctypes.data_as(ctypes.c_void_p),
                        warped.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(h), ctypes.c_int(w))
    save(name, h=h, w=w, src=src, idx=idx, idy=idy, z=z, warped=warped.reshape(h, w, 5))


def gen_exp():
    rs = np.random.RandomState(3)
    x = np.concatenate([-rs.rand(3000) * 20, -10 ** (rs.rand(3000) * 8 - 6), -rs.rand(500) * 110,
                        [0.0, -0.0, -1e-10, -87.5, -88.0, -100.0, -103.9, -104.5]]).astype(np.float32)
    y = torch.exp(torch.from_numpy(x)).numpy()
    save("exp_vectors", x=x, y=y)


def gen_pose_schedule(seed=114514, n=16, ext_cz=0.15):
    """gen_3dphoto_dynamic_v2.py:38-39 seeds once; each pair then draws a dynamic pose and a camera pose."""
    random.seed(seed)
    dyn, cam = [], []
    for _ in range(n):
        dyn.append(R.utils.generate_random_pose(ext_cz).numpy())
        cam.append(R.utils.generate_random_pose(ext_cz, base_motions=[0, 0, 0]).numpy())
    save("pose_schedule", seed=seed, ext_cz=ext_cz, G_dyn=np.stack(dyn), G_cam=np.stack(cam))
    # the two other copies of the sampler the reference ships (utils/utils_coco.py:121-154, "utils/utils copy.py":121-160)
    extra = {}
    V = ref_harness.variant_utils()
    for key, mod in (("coco", V.coco), ("copy", V.copy)):
        random.seed(seed)
        d2, c2 = [], []
        for _ in range(n):
            d2.append(mod.generate_random_pose().numpy())
            c2.append(mod.generate_random_pose(base_motions=[0, 0, 0]).numpy())
        extra["G_dyn_" + key], extra["G_cam_" + key] = np.stack(d2), np.stack(c2)
    save("pose_schedule_variants", seed=seed, **extra)


def gen_alpha(seed=23):
    """render(use_alpha=True) = alpha_composition (mpi_rendering.py:33-59) and get_xyz_from_depth (:157-177)."""
    rs = np.random.RandomState(seed)
    B, S, H, W = 1, 20, 10, 14
    sigma = rs.rand(B, S, 1, H, W).astype(np.float32)
    sigma[:, ::5] = 0.0
    sigma[:, 3] = 1.0
    rgb = rs.rand(B, S, 3, H, W).astype(np.float32)
    xyz = (rs.rand(B, S, 3, H, W).astype(np.float32) * 4 + 0.5)
    T = torch.from_numpy
    try:                                   # the reference's own use_alpha branch never binds flowA2B (:33-39): record that it raises
        R.mpi_rendering.render(T(rgb), T(sigma), T(xyz), use_alpha=True)
        render_raises = ""
    except UnboundLocalError as e:
        render_raises = type(e).__name__
    imgs, weights = R.mpi_rendering.alpha_composition(T(sigma), T(rgb))
    depth, _ = R.mpi_rendering.alpha_composition(T(sigma), T(xyz[:, :, 2:]))
    blend = torch.cumprod(1 - T(sigma) + 1e-6, dim=1)                      # :36
    dep = (rs.rand(B, 1, H, W).astype(np.float32) * 5 + 0.5)
    K = synth.intrinsics(H, W)
    k_inv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    mesh = R.homography_sampler.HomographySample(H, W, device=torch.device("cpu")).meshgrid
    xyz_d = R.mpi_rendering.get_xyz_from_depth(mesh, T(dep), T(k_inv)[None])
    save("alpha_composition", sigma=sigma, rgb=rgb, xyz=xyz, imgs=imgs.numpy(), depth=depth.numpy(), blend_weights=blend.numpy(),
         weights=weights.numpy(), depth_map=dep, K_inv=k_inv, xyz_from_depth=xyz_d.numpy(), render_use_alpha_raises=render_raises)


def gen_flo(seed=31):
    """write_flow.writeFlow (write_flow.py:74-103): the bytes of a small .flo file as the reference writes it."""
    import importlib
    import tempfile
    wf = importlib.import_module("write_flow")
    rs = np.random.RandomState(seed)
    flow = (rs.randn(7, 11, 2) * 30).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "a.flo"), os.path.join(d, "b.flo")
        wf.writeFlow(a, flow)
        wf.writeFlow(b, flow[:, :, 0].astype(np.float64), flow[:, :, 1].astype(np.float64))
        raw_a, raw_b = open(a, "rb").read(), open(b, "rb").read()
        back = wf.readFlow(a)
    save("flo_file", flow=flow, file_bytes=np.frombuffer(raw_a, np.uint8), file_bytes_uv=np.frombuffer(raw_b, np.uint8), read_back=back)


def gen_geometry(seed=11):
    """geometry.py known answers: transformation_from_parameters (both branches), BackprojectDepth, Project3D."""
    rs = np.random.RandomState(seed)
    aa = (rs.rand(6, 1, 3).astype(np.float32) - 0.5) * 0.2
    tr = (rs.rand(6, 3).astype(np.float32) - 0.5)
    g = R.geometry
    M = g.transformation_from_parameters(torch.from_numpy(aa), torch.from_numpy(tr)).numpy()
    Mi = g.transformation_from_parameters(torch.from_numpy(aa), torch.from_numpy(tr), invert=True).numpy()
    H, W = 20, 28
    depth = (rs.rand(1, H, W).astype(np.float32) * 5 + 0.5)
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = synth.intrinsics(H, W)
    iK = np.eye(4, dtype=np.float32)
    iK[:3, :3] = np.linalg.inv(K[:3, :3].astype(np.float64)).astype(np.float32)
    bp, pj = g.BackprojectDepth(1, H, W), g.Project3D(1, H, W)
    cam = bp(torch.from_numpy(depth), torch.from_numpy(iK)[None])
    pix, z = pj(cam, torch.from_numpy(K)[None], torch.from_numpy(M[:1]))
    save("geometry", axisangle=aa, translation=tr, M=M, M_inv=Mi, depth=depth, K4=K, inv_K4=iK,
         cam_points=cam.detach().numpy(), pix=pix.detach().numpy(), z=z.detach().numpy(), T=M[0])


def gen_hard_flow(name="hard_flow", S=8, H=24, W=40, seed=6, pose_seed=12):
    """render_3dphoto_dynamic(..., hard_flow=True): flow of the arg-max-weight plane (mpi_rendering.py:126-130)."""
    inp = synth.make_inputs(S, H, W, seed=seed, kind="white")
    mpi = torch.from_numpy(inp["mpi"])[None]
    random.seed(pose_seed)
    flow_mix, src_np, _, _ = R.utils.render_3dphoto_dynamic(Opt, torch.from_numpy(inp["image"])[None], torch.from_numpy(inp["obj_mask"])[None, None],
                                                            None, mpi, torch.from_numpy(inp["disparity"])[None], torch.from_numpy(inp["K"])[None],
                                                            torch.from_numpy(inp["K"])[None], name="g", hard_flow=True)
    save(name, S=S, H=H, W=W, seed=seed, pose_seed=pose_seed, flow_mix=flow_mix, obj_mask=inp["obj_mask"])


def gen_model(S=4, H=128, W=128, seed=0):
    """AdaMPI network (N1): the REFERENCE model/AdaMPI.py, loaded (strict) with the deterministic parameters that
    mpiflow_amd.model.MPIPredictor.randomize_(seed) produces, run on CPU fp32."""
    from model.AdaMPI import MPIPredictor as RefModel
    from mpiflow_amd.model import MPIPredictor
    mine = MPIPredictor(W, H, S).randomize_(seed)
    ref = RefModel(width=W, height=H, num_planes=S)
    ref.load_state_dict(mine.state_dict(), strict=True)
    ref.eval()
    g = torch.Generator().manual_seed(seed + 1)
    img, dsp = torch.rand(1, 3, H, W, generator=g), torch.rand(1, 1, H, W, generator=g)
    with torch.no_grad():
        mpi, disp = ref(img, dsp)
    save("model_adampi", S=S, H=H, W=W, seed=seed, image=img.numpy(), disp=dsp.numpy(), plane_disp=disp.numpy(),
         mpi_sub=mpi[0, :, :, ::2, ::2].numpy(), mpi_sum=np.array(mpi.double().sum().item()),
         n_state=len(ref.state_dict()), sha_keys=np.array(sha(np.frombuffer("|".join(sorted(ref.state_dict())).encode(), np.uint8))))


def gen_e2e(name="e2e_loop_body", S=8, H=128, W=256, seed=0, pose_seed=61):
    """The reference's LOOP BODY end to end (gen_3dphoto_dynamic_v2.py:82-118) in fp32 on the CPU: its own input resize (F.interpolate bilinear,
    align_corners=True, :86-89, :104-105), its own network (model/AdaMPI.py MPIPredictor with the deterministic parameters of
    mpiflow_amd.model.MPIPredictor.randomize_(seed), loaded strict - the published checkpoint is not available offline), and its own
    render_3dphoto_dynamic on the stack the network produced (:107-118).  The only fixture whose sigma field has the network's distribution
    (relu(x * cum_mask) + 1e-4, model/CPN/decoder.py:166-173) instead of synth.make_inputs' white / smooth / opaque draws.  Stored: the resized inputs,
    the network's stack, both posed views un-thresholded, the merge products and cv2.inpaint's inputs."""
    import torch.nn.functional as F
    from model.AdaMPI import MPIPredictor as RefModel
    from mpiflow_amd.model import MPIPredictor
    mine = MPIPredictor(W, H, S).randomize_(seed)
    ref = RefModel(width=W, height=H, num_planes=S)
    ref.load_state_dict(mine.state_dict(), strict=True)
    ref.eval()
    rs = np.random.RandomState(seed + 7)
    h0, w0 = 100, 230                                                     # the files' own size, resized to (H, W) as the reference does
    yy, xx = np.mgrid[0:h0, 0:w0].astype(np.float32)
    img0 = np.stack([0.5 + 0.3 * np.sin(xx / 9.0 + c) * np.cos(yy / 7.0 - c) + 0.15 * rs.rand(h0, w0) for c in range(3)]).astype(np.float32).clip(0, 1)
    dsp0 = (0.15 + 0.7 * yy / h0 + 0.1 * np.sin(xx / 23.0) + 0.03 * rs.rand(h0, w0)).astype(np.float32).clip(0, 1)
    ids = np.zeros((h0, w0), np.uint8)
    ids[30:70, 60:120] = 1
    ids[45:90, 150:200] = 2
    image = F.interpolate(torch.from_numpy(img0)[None], size=(H, W), mode="bilinear", align_corners=True)       # :86-87
    disp = F.interpolate(torch.from_numpy(dsp0)[None, None], size=(H, W), mode="bilinear", align_corners=True)  # :88-89
    with torch.no_grad():
        mpi_all_src, disparity_all_src = ref(image, disp)                                                         # :92-93
    np.random.seed(seed)
    obj_index = np.random.randint(ids.max()) + 1                                                                   # :101
    obj_mask = torch.FloatTensor(ids == obj_index).unsqueeze(0).unsqueeze(0)                                      # :103
    obj_mask = F.interpolate(obj_mask, size=(H, W), mode="bilinear", align_corners=True)                          # :104-105
    K = torch.tensor([[0.58, 0, 0.5], [0, 0.58, 0.5], [0, 0, 1]])                                                  # :42-49
    K[0, :] *= W
    K[1, :] *= H
    K = K.unsqueeze(0)
    G_cam, G_dyn = poses_for(pose_seed)
    random.seed(pose_seed)
    ref_harness.captured.clear()
    flow_mix, src_np, inpainted, _ = R.utils.render_3dphoto_dynamic(Opt, image, obj_mask, disp, mpi_all_src, disparity_all_src, K, K,
                                                                    data_path="outputs", name="demo")              # :107-118
    cap = ref_harness.captured["inpaint"]
    inp = dict(mpi=mpi_all_src[0].numpy(), disparity=disparity_all_src[0].numpy(), K=K[0].numpy(), image=image[0].numpy(), obj_mask=obj_mask[0, 0].numpy())
    views = run_views(inp, G_cam.numpy(), G_dyn.numpy())
    sig = inp["mpi"][:, 3]
    arrays = dict(S=S, H=H, W=W, seed=seed, pose_seed=pose_seed, image_file=img0, disp_file=dsp0, ids_file=ids, obj_index=obj_index,
                  mpi=inp["mpi"], disparity=inp["disparity"], image=inp["image"], disp=disp[0, 0].numpy(), obj_mask=inp["obj_mask"], K=inp["K"],
                  G_cam=G_cam.numpy(), G_dyn=G_dyn.numpy(), flow_mix=flow_mix, src_np=src_np, frame_mix=cap["img"], fill_mask=cap["mask"].astype(np.uint8),
                  sigma_floor_share=np.array(float((sig == np.float32(1e-4)).mean())), sigma_max=np.array(float(sig.max())))
    for tag in ("cam", "dyn"):
        for k, v in views[tag].items():
            arrays["%s_%s" % (tag, k)] = v
        arrays["margin_px_%s" % tag] = margin_pixels(views[tag]["objmask"])
    arrays["margin_px_obj"] = margin_pixels(inp["obj_mask"])
    print("   network stack: sigma == 1e-4 on %.1f %% of the entries, max %.3g; fill mask %d px; margin band %d + %d px" %
          (100 * float(arrays["sigma_floor_share"]), float(arrays["sigma_max"]), int(arrays["fill_mask"].astype(bool).sum()),
           arrays["margin_px_cam"].size, arrays["margin_px_dyn"].size))
    save(name, **arrays)


def gen_input_stage(seed=41):
    """The input stage as the reference runs it (gen_3dphoto_dynamic_v2.py:82-89, :101-105 with utils/utils.py:35-52):
    image_to_tensor / disparity_to_tensor on PNG files, the three F.interpolate(bilinear, align_corners=True) calls and the
    (ids == k) instance mask.  cv2.imread is third-party and absent; for an 8-BIT GREY PNG `cv2.imread(path, 0)` returns the
    stored bytes, so the stub decodes exactly that case with PIL (other encodings of a disparity file are not pinned here).
    Two output sizes: 96 x 160 (out_H + out_W > 128: ATen's generic kernel, as every real size) and 40 x 72 (<= 128: its
    channels-last kernel)."""
    import sys as _sys
    import tempfile
    import torch.nn.functional as F
    from PIL import Image
    rs = np.random.RandomState(seed)
    h, w = 90, 150
    rgb = (rs.rand(h, w, 3) * 256).astype(np.uint8)
    dsp = (synth._upsample(rs.rand(6, 10), h, w) * 255).astype(np.uint8)
    ids = np.zeros((h, w), np.uint8)
    ids[10:40, 20:70] = 1
    ids[30:80, 90:140] = 2
    ids[60:75, 10:60] = 3
    arrays = dict(rgb_u8=rgb, disp_u8=dsp, ids_u8=ids, torch_threads=torch.get_num_threads())
    _sys.modules["cv2"].imread = lambda path, flag=1: np.array(Image.open(path).convert("L"))
    with tempfile.TemporaryDirectory() as d:
        Image.fromarray(rgb).save(os.path.join(d, "i.png"))
        Image.fromarray(dsp).save(os.path.join(d, "d.png"))
        Image.fromarray(ids).save(os.path.join(d, "m.png"))
        image = R.utils.image_to_tensor(os.path.join(d, "i.png"))                          # [1,3,h,w]
        obj_mask_np = np.array(Image.open(os.path.join(d, "m.png")).convert("L"))
        disp = R.utils.disparity_to_tensor(os.path.join(d, "d.png"))                        # [1,1,h,w]
    arrays.update(image_tensor=image[0].numpy(), disp_tensor=disp[0, 0].numpy(), mask_max=int(obj_mask_np.max()))
    for tag, (H, W) in (("big", (96, 160)), ("small", (40, 72))):
        im = F.interpolate(image, size=(H, W), mode="bilinear", align_corners=True)
        dp = F.interpolate(disp, size=(H, W), mode="bilinear", align_corners=True)
        arrays["image_" + tag], arrays["disp_" + tag] = im[0].numpy(), dp[0, 0].numpy()
        for k in (1, 2, 3):
            om = torch.FloatTensor(obj_mask_np == k).unsqueeze(0).unsqueeze(0)
            om = F.interpolate(om, size=(H, W), mode="bilinear", align_corners=True)
            arrays["mask%d_%s" % (k, tag)] = om[0, 0].numpy()
    save("input_stage", **arrays)


def gen_copy_variant(name="copy_variant", cases=((8, 96, 128, "smooth", 51, 71), (12, 72, 96, "white", 52, 70))):
    """The reference's OLDER per-image module, "utils/utils copy.py" (own pose constants, :121-160; depth-ordered frame, :295-303): its
    render_3dphoto_dynamic end to end under random.seed, with what it hands to its two cv2.inpaint calls (:309-315) - the merged frame
    and the depth-ordered frame - and the two views' depths it compares."""
    import tempfile
    V = ref_harness.variant_utils().copy
    arrays = dict(n_cases=len(cases))
    for ci, (S, H, W, kind, seed, pose_seed) in enumerate(cases):
        inp = synth.make_inputs(S, H, W, seed=seed, kind=kind)
        # a large binary object (the synthetic default is a small soft box): the depth-ordered pick only changes pixels where the object
        # layer's rendered mask reaches the threshold AND the background layer's is non-zero, i.e. in a band along the object's outline
        box = np.zeros((H, W), np.float32)
        box[H // 5:4 * H // 5, W // 6:5 * W // 6] = 1.0
        inp["obj_mask"] = box
        mpi, disp = torch.from_numpy(inp["mpi"])[None], torch.from_numpy(inp["disparity"])[None]
        K, img = torch.from_numpy(inp["K"])[None], torch.from_numpy(inp["image"])[None]
        om = torch.from_numpy(inp["obj_mask"])[None, None]
        random.seed(pose_seed)
        G_dyn = V.generate_random_pose()
        G_cam = V.generate_random_pose(base_motions=[0, 0, 0])
        seen, real = [], V.render_novel_view_dynamic

        def spy(*a, **k):
            out = real(*a, **k)
            seen.append(out)
            return out
        V.render_novel_view_dynamic = spy
        random.seed(pose_seed)
        ref_harness.captured.clear()
        try:
            with tempfile.TemporaryDirectory() as d:
                flow_mix, src_np, inpainted, res = V.render_3dphoto_dynamic(img, om, None, mpi, disp, K, K, data_path=d, name="golden")
        finally:
            V.render_novel_view_dynamic = real
        calls = ref_harness.captured["inpaint_calls"]
        assert len(calls) == 2 and len(seen) == 2 and np.array_equal(calls[0]["mask"], calls[1]["mask"])
        pre = "c%d_" % ci
        arrays.update({pre + "S": S, pre + "H": H, pre + "W": W, pre + "kind": kind, pre + "seed": seed, pre + "pose_seed": pose_seed,
                       pre + "mpi": inp["mpi"], pre + "disparity": inp["disparity"], pre + "image": inp["image"], pre + "obj_mask": inp["obj_mask"],
                       pre + "K": inp["K"], pre + "G_cam": G_cam.numpy(), pre + "G_dyn": G_dyn.numpy(), pre + "flow_mix": flow_mix,
                       pre + "src_np": src_np, pre + "frame_mix": calls[0]["img"], pre + "fill_mask": calls[0]["mask"].astype(np.uint8),
                       pre + "frame_mix_depth": calls[1]["img"],
                       pre + "cam_rgb": seen[0][0][0].numpy(), pre + "cam_depth": seen[0][1][0, 0].numpy(), pre + "cam_objmask": seen[0][3][0, 0].numpy(),
                       pre + "dyn_rgb": seen[1][0][0].numpy(), pre + "dyn_depth": seen[1][1][0, 0].numpy(), pre + "dyn_objmask": seen[1][3][0, 0].numpy()})
        n_pick = int((calls[1]["img"] != calls[0]["img"]).any(-1).sum())
        print("   case %d: %d x %d x %d %s: depth-ordered frame differs from the merged one on %d px" % (ci, S, H, W, kind, n_pick))
    save(name, **arrays)


C2_OPAQUE_SEED = 111    # found with the pinned oracle (first seed of 100.. whose two rendered masks keep 2e-5 from the threshold)


JOBS = {
    "tiny": lambda: (gen_small("tiny_white", 8, 32, 48, "white", 1, 7, True),
                     gen_small("tiny_smooth", 8, 32, 48, "smooth", 2, 8, True)),
    "odd": lambda: (gen_small("odd_s20", 20, 23, 37, "white", 3, 9, False),
                    gen_small("odd_s5", 5, 17, 19, "smooth", 4, 10, False),
                    gen_small("s1", 1, 16, 24, "white", 5, 11, False)),
    "c1": lambda: gen_big("c1_white", 32, 384, 512, "white", 11, 21),
    "c2": lambda: (gen_big("c2_white", 64, 640, 960, "white", 12, 22), gen_big("c2_smooth", 64, 640, 960, "smooth", 13, 23)),
    # BASELINE configs[4]: 128 planes, reference sampler's random poses.  The reference itself needs > 58 GB at the full
    # 1024 x 1536 (it was tried here: allocation failure under a 58 GB limit on this 62 GB container), so the golden is recorded
    # at 128 x 512 x 768 - same plane count (the 128-addend cascade sums), a quarter of the pixels; the full-size frame is
    # covered by HIP-vs-pinned-oracle on every pixel (tests/test_full_frame.py).
    "c5": lambda: gen_big("c5q_white", 128, 512, 768, "white", 14, 24, stack_px=1024),
    # the generator's real shape (gen_3dphoto_dynamic_v2.py:22-23 defaults: 64 planes, 384 x 1280 - KITTI)
    "kitti": lambda: (gen_big("kitti_smooth", 64, 384, 1280, "smooth", 15, 25, stack_px=2048), gen_big("kitti_white", 64, 384, 1280, "white", 16, 26, stack_px=2048)),
    # SURVEY section 7, hard part 2: inputs whose margin band (rendered-mask values within 1e-5 of 0.99) is EMPTY, so the thresholded
    # masks compare bit for bit on every pixel without exclusions (synth kind '*_opaque': opaque last plane, small object; seeds found by
    # searching with the pinned oracle, and asserted here on the reference's own values at twice the margin)
    "opaque": lambda: (gen_big("c1_opaque", 32, 384, 512, "smooth_opaque", 32, 42, require_empty_margin=True),
                       gen_big("kitti_opaque", 64, 384, 1280, "smooth_opaque", 40, 50, stack_px=2048, require_empty_margin=True),
                       gen_big("c2_opaque", 64, 640, 960, "smooth_opaque", C2_OPAQUE_SEED, C2_OPAQUE_SEED + 10, require_empty_margin=True)),
    "fwarp": lambda: (gen_fwarp("fwarp_small", 96, 128, 31, True), gen_fwarp("fwarp_c2", 640, 960, 32, False),
                      gen_collision_stress()),
    "inputs": gen_input_stage,
    "exp": gen_exp,
    "pose": gen_pose_schedule,
    "alpha": gen_alpha,
    "flo": gen_flo,
    "geometry": gen_geometry,
    "model": gen_model,
    "e2e": gen_e2e,
    "hard": gen_hard_flow,
    "copy": gen_copy_variant,
}

if __name__ == "__main__":
    import subprocess
    subprocess.run(["make", "-C", os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"), "ref"], check=True)
    todo = sys.argv[1:] or list(JOBS)
    for j in todo:
        print("== %s" % j)
        JOBS[j]()
