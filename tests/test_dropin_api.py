"""GPU tests of the drop-in layer: the reference's own function signatures (utils/utils.py, utils/mpi/*, geometry.py,
moving_obj.py), called the way the reference calls them, checked against tensors recorded from the reference.

Everything that does not pass through exp is asserted bit-exact; composites within 2e-6; flow within 5e-5."""
import random

import numpy as np
import pytest
import torch

from conftest import bits_equal, load_golden, max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mpiflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", ["tiny_white", "tiny_smooth"])
def test_homography_sampler_and_xyz_functions(dev, name):
    from mpiflow_amd.utils.mpi import mpi_rendering
    from mpiflow_amd.utils.mpi.homography_sampler import HomographySample
    from mpiflow_amd.utils.mpi.rendering_utils import transform_G_xyz
    g = load_golden(name)
    S, H, W = int(g["S"]), int(g["H"]), int(g["W"])
    hs = HomographySample(H, W, dev)
    assert bits_equal(N(hs.meshgrid), g["meshgrid"]) == 0 and hs.Height_tgt == H and hs.Width_tgt == W
    assert N(hs.n).tolist() == [0, 0, 1]
    disp = T(g["disparity"], dev)[None]
    K = T(g["K"], dev)[None]
    k_inv = T(g["k_inv"], dev)[None]
    G = T(g["G_cam"], dev)
    xyz_src = mpi_rendering.get_src_xyz_from_plane_disparity(hs.meshgrid, disp, k_inv)
    assert tuple(xyz_src.shape) == (1, S, 3, H, W) and bits_equal(N(xyz_src[0]), g["xyz_src"]) == 0
    xyz_tgt = mpi_rendering.get_tgt_xyz_from_plane_disparity(xyz_src, G[None])
    assert bits_equal(N(xyz_tgt[0]), g["xyz_tgt_cam"]) == 0
    out = transform_G_xyz(G[None].repeat(S, 1, 1), xyz_src[0].reshape(S, 3, H * W))
    assert bits_equal(N(out).reshape(S, 3, H, W), g["xyz_tgt_cam"]) == 0
    # HomographySample.sample / sample_inverse exactly as render_tgt_rgb_depth calls them
    om = T(g["obj_mask"], dev)[None, None].repeat(S, 1, 1, 1)
    cat = torch.cat((T(g["rgb_blended"], dev), T(g["mpi"][:, 3:], dev), xyz_tgt[0], om), dim=1)
    d = torch.reciprocal(disp)[0]
    GS, KiS, KS = G[None].repeat(S, 1, 1), k_inv.repeat(S, 1, 1), K.repeat(S, 1, 1)
    tgt, valid, fB2A = hs.sample(cat, d, GS, KiS, KS)
    assert valid.dtype == torch.bool
    assert bits_equal(N(tgt), g["sample_tgt"]) == 0
    assert bits_equal(N(valid), g["sample_valid"]) == 0
    assert bits_equal(N(fB2A), g["sample_flowB2A"]) == 0
    fA2B = hs.sample_inverse(cat, d, GS, KiS, KS)
    assert bits_equal(N(fA2B), g["sample_inverse_flow"]) == 0


@pytest.mark.parametrize("name", ["tiny_white", "tiny_smooth"])
def test_render_and_render_tgt_rgb_depth(dev, name):
    from mpiflow_amd.utils.mpi import mpi_rendering
    from mpiflow_amd.utils.mpi.homography_sampler import HomographySample
    g = load_golden(name)
    S, H, W = int(g["S"]), int(g["H"]), int(g["W"])
    hs = HomographySample(H, W, dev)
    mpi = T(g["mpi"], dev)[None]
    rgb, sig = mpi[:, :, 0:3], mpi[:, :, 3:]
    xyz_src = T(g["xyz_src"], dev)[None]
    imgs, depth, bw, w, flow, om = mpi_rendering.render(rgb, sig, xyz_src, use_alpha=False, is_bg_depth_inf=False)
    assert flow is None and om is None
    assert tuple(bw.shape) == (1, S, 1, H, W) and tuple(imgs.shape) == (1, 3, H, W) and tuple(depth.shape) == (1, 1, H, W)
    assert max_abs(N(bw[0, :, 0]), g["blend_weights"]) < 1e-6
    assert max_abs(N(w[0, :, 0]), g["weights_src"]) < 1e-6
    # weighted_sum_mpi on the reference's own weights is exp-free -> bit exact vs a cascade of the same products
    r2, d2 = mpi_rendering.weighted_sum_mpi(rgb, xyz_src, T(g["weights_src"], dev)[None, :, None], False)
    assert tuple(r2.shape) == (1, 3, H, W) and tuple(d2.shape) == (1, 1, H, W)
    rgb_b = T(g["rgb_blended"], dev)[None]
    xyz_tgt = T(g["xyz_tgt_cam"], dev)[None]
    om_in = T(g["obj_mask"], dev)[None, None, None].repeat(1, S, 1, 1, 1)
    out = mpi_rendering.render_tgt_rgb_depth(hs, rgb_b, sig, T(g["disparity"], dev)[None], xyz_tgt, xyz_src, T(g["G_cam"], dev)[None],
                                             T(g["k_inv"], dev)[None], T(g["K"], dev)[None], None, obj_mask=om_in)
    r_rgb, r_depth, r_tmask, r_flow, r_om = out
    assert max_abs(N(r_rgb[0]), g["rtd_rgb"]) < 2e-6
    assert max_abs(N(r_om[0, 0]), g["rtd_objmask"]) < 2e-6
    assert bits_equal(N(r_tmask[0, 0]), g["rtd_tgt_mask"]) == 0
    assert max_abs(N(r_flow[0]), g["rtd_flow_unclipped"]) < 5e-5
    assert max_abs(N(r_depth[0, 0]), g["rtd_depth"]) < 2e-5 * max(1.0, float(np.abs(g["rtd_depth"]).max()))
    with pytest.raises(UnboundLocalError):        # what the reference's own use_alpha branch does (mpi_rendering.py:33-39)
        mpi_rendering.render(rgb, sig, xyz_src, use_alpha=True)
    # xyz tensors loaded from a file carry no provenance tag: that call took the GENERIC kernels
    assert not mpi_rendering.fused_render_applies(T(g["disparity"], dev)[None], xyz_tgt, xyz_src, T(g["G_cam"], dev)[None], T(g["k_inv"], dev)[None],
                                                  T(g["K"], dev)[None], obj_mask=om_in)
    with pytest.raises(ValueError):
        mpi_rendering.render_tgt_rgb_depth(hs, rgb_b, sig, T(g["disparity"], dev)[None], xyz_tgt, xyz_src, T(g["G_cam"], dev)[None],
                                           T(g["k_inv"], dev)[None], T(g["K"], dev)[None], None, obj_mask=om_in, fused=True)
    # the reference's own call sequence (utils/utils.py:303-349): xyz tensors from this module's two functions -> the FUSED kernels
    # (mpf_warp_composite_split + mpf_src_flow).  Same bars against the reference's outputs; tgt_mask identical; against the generic form the
    # difference is fp32 rounding of the xyz channels (interpolated there, evaluated at the interpolated coordinate here)
    disp, G, Ki, K = T(g["disparity"], dev)[None], T(g["G_cam"], dev)[None], T(g["k_inv"], dev)[None], T(g["K"], dev)[None]
    xs = mpi_rendering.get_src_xyz_from_plane_disparity(hs.meshgrid, disp, Ki)
    xt = mpi_rendering.get_tgt_xyz_from_plane_disparity(xs.to(K.dtype), G.to(K.dtype))
    assert bits_equal(N(xs[0]), g["xyz_src"]) == 0 and bits_equal(N(xt[0]), g["xyz_tgt_cam"]) == 0
    assert mpi_rendering.fused_render_applies(disp, xt, xs, G, Ki, K, obj_mask=om_in)
    f_rgb, f_depth, f_tmask, f_flow, f_om = mpi_rendering.render_tgt_rgb_depth(hs, rgb_b, sig, disp, xt, xs, G, Ki, K, None, obj_mask=om_in)
    assert tuple(f_rgb.shape) == (1, 3, H, W) and tuple(f_depth.shape) == (1, 1, H, W) and tuple(f_tmask.shape) == (1, 1, H, W)
    assert tuple(f_flow.shape) == (1, 2, H, W) and tuple(f_om.shape) == (1, 1, H, W)
    assert max_abs(N(f_rgb[0]), g["rtd_rgb"]) < 2e-6 and max_abs(N(f_om[0, 0]), g["rtd_objmask"]) < 2e-6
    assert bits_equal(N(f_tmask[0, 0]), g["rtd_tgt_mask"]) == 0 and torch.equal(f_tmask, r_tmask)
    assert max_abs(N(f_flow[0]), g["rtd_flow_unclipped"]) < 5e-5
    assert max_abs(N(f_depth[0, 0]), g["rtd_depth"]) < 2e-5 * max(1.0, float(np.abs(g["rtd_depth"]).max()))
    assert max_abs(N(f_rgb), N(r_rgb)) < 1e-6 and max_abs(N(f_flow), N(r_flow)) < 1e-5
    gen = mpi_rendering.render_tgt_rgb_depth(hs, rgb_b, sig, disp, xt, xs, G, Ki, K, None, obj_mask=om_in, fused=False)
    assert all(torch.equal(a, b) for a, b in zip(gen, out))                      # fused=False: the generic kernels, whatever the tags say
    # what must NOT dispatch: another pose than the one the tensor was built for, a tensor modified in place, per-plane masks, non-default options
    assert not mpi_rendering.fused_render_applies(disp, xt, xs, T(g["G_dyn"], dev)[None], Ki, K)
    assert not mpi_rendering.fused_render_applies(disp, xt, xs, G, Ki, K, hard_flow=True)
    om_diff = om_in.clone()
    om_diff[0, 1] += 0.5
    assert not mpi_rendering.fused_render_applies(disp, xt, xs, G, Ki, K, obj_mask=om_diff)
    assert mpi_rendering.fused_render_applies(disp, xt, xs, G, Ki, K, obj_mask=T(g["obj_mask"], dev)[None, None, None].expand(1, S, 1, H, W))
    xt.mul_(1.0)
    assert not mpi_rendering.fused_render_applies(disp, xt, xs, G, Ki, K)
    no_mask = mpi_rendering.render_tgt_rgb_depth(hs, rgb_b, sig, disp, mpi_rendering.get_tgt_xyz_from_plane_disparity(xs, G), xs, G, Ki, K)
    assert no_mask[4] is None and max_abs(N(no_mask[0][0]), g["rtd_rgb"]) < 2e-6
    # a batch of two (the functions are batched although the entry point is batch-1): item 0 = the case above, item 1 = the other pose
    G2 = torch.cat([G, T(g["G_dyn"], dev)[None]])
    rep2 = lambda t: t.repeat(2, *([1] * (t.dim() - 1)))            # noqa: E731
    xs2 = mpi_rendering.get_src_xyz_from_plane_disparity(hs.meshgrid, rep2(disp), rep2(Ki))
    xt2 = mpi_rendering.get_tgt_xyz_from_plane_disparity(xs2, G2)
    assert mpi_rendering.fused_render_applies(rep2(disp), xt2, xs2, G2, rep2(Ki), rep2(K), obj_mask=rep2(om_in))
    fb = mpi_rendering.render_tgt_rgb_depth(hs, rep2(rgb_b), rep2(sig), rep2(disp), xt2, xs2, G2, rep2(Ki), rep2(K), None, obj_mask=rep2(om_in))
    gb = mpi_rendering.render_tgt_rgb_depth(hs, rep2(rgb_b), rep2(sig), rep2(disp), xt2, xs2, G2, rep2(Ki), rep2(K), None, obj_mask=rep2(om_in), fused=False)
    assert all(tuple(t.shape)[0] == 2 for t in fb) and torch.equal(fb[0][0], f_rgb[0]) and torch.equal(fb[3][0], f_flow[0])
    assert torch.equal(fb[2], gb[2]) and max_abs(N(fb[0]), N(gb[0])) < 1e-6 and max_abs(N(fb[3]), N(gb[3])) < 1e-5 and max_abs(N(fb[4]), N(gb[4])) < 1e-6


@pytest.mark.parametrize("name", ["tiny_white", "odd_s20", "s1"])
def test_render_novel_view_dynamic_signature(dev, name):
    from mpiflow_amd.utils import utils as U
    from mpiflow_amd.utils.mpi.homography_sampler import HomographySample
    g = load_golden(name)
    S, H, W = int(g["S"]), int(g["H"]), int(g["W"])
    hs = HomographySample(H, W, dev)
    rgb_b = T(g["rgb_blended"], dev)[None]
    sig = T(g["mpi"], dev)[None][:, :, 3:]            # a strided view, as in the reference (utils/utils.py:189)
    om = T(g["obj_mask"], dev)[None, None]
    frame, depth, flow, mask = U.render_novel_view_dynamic(om, rgb_b, sig, T(g["disparity"], dev)[None], T(g["G_cam"], dev),
                                                           T(g["k_inv"], dev)[None], T(g["K"], dev)[None], T(g["K"], dev)[None], None, hs)
    assert tuple(frame.shape) == (1, 3, H, W) and tuple(depth.shape) == (1, 1, H, W) and tuple(flow.shape) == (1, 2, H, W)
    assert tuple(mask.shape) == (1, 1, H, W)
    assert max_abs(N(frame[0]), g["cam_rgb"]) < 2e-6
    assert max_abs(N(mask[0, 0]), g["cam_objmask"]) < 2e-6
    assert max_abs(N(flow[0]), g["cam_flow"]) < 5e-5
    f2, _, fl2, m2 = U.render_novel_view_dynamic(1 - om, rgb_b, sig, T(g["disparity"], dev)[None], T(g["G_dyn"], dev),
                                                 T(g["k_inv"], dev)[None], T(g["K"], dev)[None], T(g["K"], dev)[None], None, hs)
    assert max_abs(N(f2[0]), g["dyn_rgb"]) < 2e-6 and max_abs(N(m2[0, 0]), g["dyn_objmask"]) < 2e-6
    assert max_abs(N(fl2[0]), g["dyn_flow"]) < 5e-5


@pytest.mark.parametrize("name", ["tiny_white", "tiny_smooth", "odd_s5"])
def test_render_3dphoto_dynamic_entry_point(dev, name):
    """Same call as gen_3dphoto_dynamic_v2.py:107-118, poses drawn from `random` in the reference's order."""
    from mpiflow_amd.utils import utils as U
    g = load_golden(name)

    class Opt:
        ext_cz = 0.15

    random.seed(int(g["pose_seed"]))
    K = T(g["K"], dev)[None]
    flow_mix, src_np, inpainted, res, inter = U.render_3dphoto_dynamic(
        Opt, T(g["image"], dev)[None], T(g["obj_mask"], dev)[None, None], None, T(g["mpi"], dev)[None], T(g["disparity"], dev)[None],
        K, K, data_path="outputs", name="demo.png", inpaint="hip", return_intermediates=True)
    assert res is None and isinstance(flow_mix, np.ndarray) and flow_mix.dtype == np.float32 and src_np.dtype == np.uint8
    assert bits_equal(src_np, g["src_np"]) == 0
    margin = np.zeros(g["fill_mask"].size, bool)
    margin[g["margin_px_cam"]] = True
    margin[g["margin_px_dyn"]] = True
    assert ((N(inter["fill_mask"]).ravel() != g["fill_mask"].ravel()) & ~margin).sum() == 0
    ok = ~margin
    assert max_abs(flow_mix.reshape(-1, 2)[ok], g["flow_mix"].reshape(-1, 2)[ok]) < 1e-4
    dfr = np.abs(N(inter["frame_mix"]).reshape(-1, 3)[ok].astype(np.int32) - g["frame_mix"].reshape(-1, 3)[ok].astype(np.int32))
    assert dfr.max() <= 1
    # hole filling (row A13, parity unpinned): outside the holes the frame is untouched, inside every pixel got a value
    hole = N(inter["fill_mask"]).astype(bool)
    assert (inpainted[~hole] == N(inter["frame_mix"])[~hole]).all()
    assert inpainted.shape == g["frame_mix"].shape and inpainted.dtype == np.uint8


def test_geometry_classes(dev):
    from mpiflow_amd import geometry
    g = load_golden("geometry")
    H, W = g["depth"].shape[-2:]
    bp, pj = geometry.BackprojectDepth(1, H, W), geometry.Project3D(1, H, W)
    cam = bp(T(g["depth"], dev), T(g["inv_K4"], dev)[None])
    assert tuple(cam.shape) == (1, 4, H * W) and bits_equal(N(cam), g["cam_points"]) == 0
    pix, z = pj(cam, T(g["K4"], dev)[None], T(g["T"], dev)[None])
    assert tuple(pix.shape) == (1, H, W, 2) and tuple(z.shape) == (1, 1, H * W)
    assert bits_equal(N(pix), g["pix"]) == 0 and bits_equal(N(z), g["z"]) == 0
    M = geometry.transformation_from_parameters(T(g["axisangle"], dev).cpu(), T(g["translation"], dev).cpu())
    assert bits_equal(N(M), g["M"]) == 0


def test_moveing_object_with_mask(dev, tmp_path, monkeypatch):
    from mpiflow_amd import moving_obj
    g = load_golden("fwarp_small")
    args = (None, T(g["disp"], dev)[None, None], g["rgb"].astype(np.float32), torch.from_numpy(g["K"]), torch.from_numpy(g["inv_K"]),
            T(g["inst"], dev)[None, None])
    # called exactly as the reference calls it (moving_obj.py:16 signature, positional): returns None, its product is temp/res-%06d.png
    monkeypatch.chdir(tmp_path)
    random.seed(int(g["seed"]))
    assert moving_obj.moveing_object_with_mask(*args, 7, inpaint="hip") is None
    from PIL import Image
    h, w = g["rgb"].shape[:2]
    assert Image.open(tmp_path / "temp" / "res-000007.png").size == (w, 4 * h)          # rgb | inpainted | forward-warped | validity mask
    out = moving_obj.moveing_object_with_mask(*args, 0, T_obj=torch.from_numpy(g["T_obj"])[None], inpaint="hip", write_debug_png=False,
                                              return_intermediates=True)
    assert bits_equal(N(out["safe_x"]), g["safe_x"]) == 0 and bits_equal(N(out["safe_y"]), g["safe_y"]) == 0
    assert bits_equal(N(out["z1"]), g["z1"]) == 0
    assert bits_equal(N(out["warped"]), g["warped"]) == 0
    assert bits_equal((1 - N(out["masks"]["H"])).astype(np.uint8), g["inpaint_mask"].astype(np.uint8)) == 0
    hole = g["inpaint_mask"].astype(bool)
    assert (N(out["im1"])[~hole] == g["warped"][..., :3][~hole]).all()
    # same RNG stream as the reference: drawing the object pose from `random` reproduces the recorded translation
    random.seed(int(g["seed"]))
    Ti = moving_obj.object_pose()
    assert bits_equal(Ti[0].numpy(), g["T_obj"]) == 0


def test_cli_end_to_end(dev, tmp_path):
    """gen_3dphoto_dynamic.py on a two-image synthetic dataset: files, formats, determinism under a fixed seed."""
    import subprocess, sys, os
    from PIL import Image
    from mpiflow_amd import io_formats
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = tmp_path / "data"
    for d in ("images", "disps", "masks"):
        (base / d).mkdir(parents=True)
    rs = np.random.RandomState(0)
    for i in range(2):
        Image.fromarray((rs.rand(40, 56, 3) * 255).astype(np.uint8)).save(base / "images" / ("im%d.png" % i))
        yy, xx = np.mgrid[0:40, 0:56]
        Image.fromarray((255 * (0.2 + 0.6 * xx / 56)).astype(np.uint8)).save(base / "disps" / ("im%d.png" % i))
        m = np.zeros((40, 56), np.uint8); m[10:25, 15:35] = 1; m[28:36, 5:20] = 2
        Image.fromarray(m).save(base / "masks" / ("im%d.png" % i))
    outs = []
    for run in range(2):
        out = tmp_path / ("out%d" % run)
        r = subprocess.run([sys.executable, os.path.join(root, "gen_3dphoto_dynamic.py"), "--base", str(base), "--out", str(out), "--width", "64",
                            "--height", "48", "--repeat", "2", "--planes", "16", "--inpaint", "hip", "--mpi-from", "disparity"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(out)
    for sub, ext in (("flows", "flo"), ("dst_images", "png"), ("src_images", "png")):
        files = sorted(os.listdir(outs[0] / sub))
        assert files == ["im0_0." + ext, "im0_1." + ext, "im1_0." + ext, "im1_1." + ext]
        for f in files:
            assert open(outs[0] / sub / f, "rb").read() == open(outs[1] / sub / f, "rb").read(), "non-deterministic output"
    flow = io_formats.read_flo(str(outs[0] / "flows" / "im0_0.flo"))
    assert flow.shape == (48, 64, 2) and np.isfinite(flow).all() and float(np.abs(flow).max()) > 0.1
    assert Image.open(outs[0] / "dst_images" / "im0_0.png").size == (64, 48)


def test_cli_two_ranks_write_the_files_of_one_rank(dev, tmp_path):
    """Image sharding: two ranks (gloo, both on GPU 0 - the box has one) produce byte-identical files to a
    single-rank run, because every rank replays the whole RNG schedule and each image is owned by exactly one rank."""
    import subprocess, sys, os
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = tmp_path / "data"
    for d in ("images", "disps", "masks"):
        (base / d).mkdir(parents=True)
    rs = np.random.RandomState(3)
    for i in range(3):
        Image.fromarray((rs.rand(40, 56, 3) * 255).astype(np.uint8)).save(base / "images" / ("im%d.png" % i))
        yy, xx = np.mgrid[0:40, 0:56]
        Image.fromarray((255 * (0.2 + 0.6 * xx / 56)).astype(np.uint8)).save(base / "disps" / ("im%d.png" % i))
        m = np.zeros((40, 56), np.uint8); m[10:25, 15:35] = 1; m[28:36, 5:20] = 2 + (i % 2)
        Image.fromarray(m).save(base / "masks" / ("im%d.png" % i))
    args = ["--base", str(base), "--width", "64", "--height", "48", "--repeat", "2", "--planes", "16", "--inpaint", "hip", "--mpi-from", "disparity"]
    one, two = tmp_path / "one", tmp_path / "two"
    lanes = tmp_path / "lanes"                      # two images in flight on one rank: same files again
    r = subprocess.run([sys.executable, os.path.join(root, "gen_3dphoto_dynamic.py"), "--out", str(lanes), "--lanes", "2"] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([sys.executable, os.path.join(root, "gen_3dphoto_dynamic.py"), "--out", str(one)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, MPIFLOW_DIST_BACKEND="gloo", MPIFLOW_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(root, "gen_3dphoto_dynamic.py"), "--out", str(two)] + args,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "pairs 6 " in r.stdout and "(2 ranks)" in r.stdout
    for sub in ("flows", "dst_images", "src_images"):
        files = sorted(os.listdir(one / sub))
        assert len(files) == 6 and files == sorted(os.listdir(two / sub))
        for f in files:
            assert open(one / sub / f, "rb").read() == open(two / sub / f, "rb").read(), "%s/%s differs between 1 and 2 ranks" % (sub, f)
            assert open(one / sub / f, "rb").read() == open(lanes / sub / f, "rb").read(), "%s/%s differs between 1 and 2 lanes" % (sub, f)


def _toy_dataset(base, names, empty_mask=(), corrupt=()):
    from PIL import Image
    for d in ("images", "disps", "masks"):
        (base / d).mkdir(parents=True, exist_ok=True)
    for n in names:
        rs = np.random.RandomState(sum(map(ord, n)))              # content depends on the name only
        Image.fromarray((rs.rand(40, 56, 3) * 255).astype(np.uint8)).save(base / "images" / (n + ".png"))
        yy, xx = np.mgrid[0:40, 0:56]
        Image.fromarray((255 * (0.2 + 0.6 * xx / 56)).astype(np.uint8)).save(base / "disps" / (n + ".png"))
        m = np.zeros((40, 56), np.uint8)
        if n not in empty_mask:
            m[10:25, 15:35] = 1
            m[28:36, 5:20] = 2
        Image.fromarray(m).save(base / "masks" / (n + ".png"))
        if n in corrupt:
            (base / "images" / (n + ".png")).write_bytes(b"\x89PNG this is not an image")


def _run_cli(root, args, env=None, nproc=1, port=29551):
    import subprocess, sys, os
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    return subprocess.run(cmd + [os.path.join(root, "gen_3dphoto_dynamic.py")] + args, capture_output=True, text=True, timeout=900, env=env)


def test_cli_eight_ranks_write_the_files_of_one_rank(dev, tmp_path):
    """`--gpus 8` from a bare python (what a user of an 8-GPU node types): the CLI starts eight ranks itself (gloo + the one device of this
    box), images are sharded i % 8 (some ranks own two images, some one, the listing has a corrupt picture and a mask without instances),
    rank 0 broadcasts the mask-maximum table, the skip lists are gathered, the statistics all-reduced - and every file equals the 1-rank run's."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ["im%02d" % i for i in range(11)]
    base = tmp_path / "data"
    _toy_dataset(base, names, empty_mask=("im03",), corrupt=("im06",))
    common = ["--base", str(base), "--width", "64", "--height", "48", "--repeat", "2", "--planes", "16", "--inpaint", "none", "--mpi-from", "disparity"]
    one, eight = tmp_path / "one", tmp_path / "eight"
    r1 = _run_cli(root, ["--out", str(one)] + common)
    assert r1.returncode == 0, r1.stderr[-3000:]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MPIFLOW_DIST_BACKEND="gloo", MPIFLOW_FORCE_DEVICE="0")
    r8 = _run_cli(root, ["--out", str(eight), "--gpus", "8"] + common, env=env)
    assert r8.returncode == 0, r8.stderr[-3000:]
    assert "pairs 18 " in r8.stdout and "(8 ranks)" in r8.stdout and "pairs 18 " in r1.stdout
    assert open(one / "skipped.txt").read() == open(eight / "skipped.txt").read() and "im03" in open(eight / "skipped.txt").read()
    for sub in ("flows", "dst_images", "src_images"):
        files = sorted(os.listdir(one / sub))
        assert len(files) == 18 and files == sorted(os.listdir(eight / sub))
        for f in files:
            assert open(one / sub / f, "rb").read() == open(eight / sub / f, "rb").read(), "%s/%s differs between 1 and 8 ranks" % (sub, f)


def test_cli_skips_bad_images_without_disturbing_the_others(dev, tmp_path):
    """Failure isolation (the reference dies on a mask without instances, gen_3dphoto_dynamic_v2.py:101): an image whose mask holds no
    instance consumes NO draws, a corrupt picture is skipped after its draws - either way every other image gets the files it would
    get if the bad ones were not in the listing / were fine; out/skipped.txt lists them; one rank and two ranks agree."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--width", "64", "--height", "48", "--repeat", "2", "--planes", "16", "--inpaint", "none", "--mpi-from", "disparity"]
    _toy_dataset(tmp_path / "full", ["a", "b", "c", "d"], empty_mask=("b",), corrupt=("c",))
    _toy_dataset(tmp_path / "clean", ["a", "c", "d"])                                   # b absent; c intact (its draws are consumed either way)
    r = _run_cli(root, ["--base", str(tmp_path / "full"), "--out", str(tmp_path / "o_full")] + common)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "skipped 2 image(s)" in r.stdout
    listed = dict(l.split("\t", 1) for l in open(tmp_path / "o_full" / "skipped.txt").read().splitlines())
    assert set(listed) == {"b", "c"} and "no instance" in listed["b"] and listed["c"].startswith("input:")
    r = _run_cli(root, ["--base", str(tmp_path / "clean"), "--out", str(tmp_path / "o_clean")] + common)
    assert r.returncode == 0, r.stderr[-3000:]
    for sub, ext in (("flows", "flo"), ("dst_images", "png"), ("src_images", "png")):
        assert sorted(os.listdir(tmp_path / "o_full" / sub)) == ["%s_%d.%s" % (n, k, ext) for n in ("a", "d") for k in (0, 1)]
        for n in ("a", "d"):
            for k in (0, 1):
                f = "%s_%d.%s" % (n, k, ext)
                assert open(tmp_path / "o_full" / sub / f, "rb").read() == open(tmp_path / "o_clean" / sub / f, "rb").read(), f
    env = dict(os.environ, MPIFLOW_DIST_BACKEND="gloo", MPIFLOW_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    r = _run_cli(root, ["--base", str(tmp_path / "full"), "--out", str(tmp_path / "o_two")] + common, env=env, nproc=2, port=29552)
    assert r.returncode == 0, r.stderr[-3000:]
    assert sorted(open(tmp_path / "o_two" / "skipped.txt").read().splitlines()) == sorted(open(tmp_path / "o_full" / "skipped.txt").read().splitlines())
    for sub in ("flows", "dst_images", "src_images"):
        for f in os.listdir(tmp_path / "o_full" / sub):
            assert open(tmp_path / "o_full" / sub / f, "rb").read() == open(tmp_path / "o_two" / sub / f, "rb").read(), f


def test_cli_gpus_flag_starts_the_ranks_and_inputs_of_mixed_sizes_render(dev, tmp_path):
    """(1) `gen_3dphoto_dynamic.py --gpus 2` without a launcher starts its two ranks itself (gloo + one device here) and writes the files
    of a one-rank run.  (2) A disparity map saved at another resolution than its image is resized on its own, as the reference does
    (gen_3dphoto_dynamic_v2.py:86-89): the image renders instead of landing in skipped.txt, and the result is that of handing
    mpf_prepare_inputs the arrays one by one."""
    import os
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--width", "64", "--height", "48", "--repeat", "2", "--planes", "16", "--inpaint", "none", "--mpi-from", "disparity"]
    base = tmp_path / "d"
    _toy_dataset(base, ["a", "b", "c"])
    yy, xx = np.mgrid[0:24, 0:32]
    Image.fromarray((255 * (0.2 + 0.6 * xx / 32)).astype(np.uint8)).save(base / "disps" / "b.png")        # b: disparity at 24 x 32, image 40 x 56
    r = _run_cli(root, ["--base", str(base), "--out", str(tmp_path / "one")] + common)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(tmp_path / "one" / "skipped.txt").read() == "" and "pairs 6 " in r.stdout
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MPIFLOW_DIST_BACKEND="gloo", MPIFLOW_FORCE_DEVICE="0")
    r = _run_cli(root, ["--base", str(base), "--out", str(tmp_path / "two"), "--gpus", "2"] + common, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "pairs 6 " in r.stdout and "(2 ranks)" in r.stdout
    for sub in ("flows", "dst_images", "src_images"):
        files = sorted(os.listdir(tmp_path / "one" / sub))
        assert len(files) == 6 and files == sorted(os.listdir(tmp_path / "two" / sub))
        for f in files:
            assert open(tmp_path / "one" / sub / f, "rb").read() == open(tmp_path / "two" / sub / f, "rb").read(), f
    # a launcher whose rank count disagrees with --gpus is an error
    r = _run_cli(root, ["--base", str(base), "--out", str(tmp_path / "x"), "--gpus", "3"] + common, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "must agree" in r.stderr


def test_cli_resume_and_defaults(dev, tmp_path):
    """--resume re-renders only what is missing and reproduces the same bytes; the default producer is the network, and a missing
    checkpoint is an error, not a silent switch to another data generator."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _toy_dataset(tmp_path / "data", ["a", "b", "c"])
    common = ["--base", str(tmp_path / "data"), "--width", "64", "--height", "48", "--repeat", "2", "--planes", "16", "--mpi-from", "disparity"]
    out = tmp_path / "out"
    r = _run_cli(root, common + ["--out", str(out)])
    assert r.returncode == 0, r.stderr[-3000:]
    want = {(sub, f): open(out / sub / f, "rb").read() for sub in ("flows", "dst_images", "src_images") for f in os.listdir(out / sub)}
    assert len(want) == 18
    os.remove(out / "flows" / "b_1.flo")                                   # image b is incomplete now
    stamp = os.path.getmtime(out / "flows" / "a_0.flo")
    r = _run_cli(root, common + ["--out", str(out), "--resume"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "resume: 2 image(s) already complete" in r.stdout and "pairs 2 " in r.stdout
    assert os.path.getmtime(out / "flows" / "a_0.flo") == stamp            # untouched
    for (sub, f), data in want.items():
        assert open(out / sub / f, "rb").read() == data, (sub, f)
    r = _run_cli(root, ["--base", str(tmp_path / "data"), "--out", str(tmp_path / "o2"), "--width", "64", "--height", "48"])
    assert r.returncode != 0 and "checkpoint" in (r.stderr + r.stdout) and "--mpi-from" in (r.stderr + r.stdout)


def test_cli_builtin_inpaint_on_the_writer_threads(dev, tmp_path, oracle):
    """--inpaint builtin: the frame leaves the GPU unfilled and a writer thread runs the restated cv2.inpaint (NS, radius 3) before
    encoding.  The written frame equals the oracle's NS fill of the --inpaint none frame, with the holes being the white pixels the
    merge left (utils/utils.py:273-286)."""
    import os
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _toy_dataset(tmp_path / "data", ["a"])
    common = ["--base", str(tmp_path / "data"), "--width", "64", "--height", "48", "--repeat", "1", "--planes", "16", "--mpi-from", "disparity"]
    for mode in ("none", "builtin"):
        r = _run_cli(root, common + ["--out", str(tmp_path / mode), "--inpaint", mode])
        assert r.returncode == 0, r.stderr[-3000:]
    raw = np.array(Image.open(tmp_path / "none" / "dst_images" / "a_0.png"))[:, :, ::-1].copy()          # file is RGB, the frame BGR
    filled = np.array(Image.open(tmp_path / "builtin" / "dst_images" / "a_0.png"))[:, :, ::-1].copy()
    hole = (raw != filled).any(-1)
    assert hole.sum() > 0 and (raw[hole] == 255).all()                     # only white (unrendered) pixels were changed
    assert open(tmp_path / "none" / "flows" / "a_0.flo", "rb").read() == open(tmp_path / "builtin" / "flows" / "a_0.flo", "rb").read()


def test_cli_with_network_producer(dev, tmp_path):
    """--mpi-from model with deterministic random weights: the network's raw output goes through the fused epilogue."""
    import subprocess, sys, os
    from PIL import Image
    from mpiflow_amd import io_formats
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = tmp_path / "data"
    for d in ("images", "disps", "masks"):
        (base / d).mkdir(parents=True)
    rs = np.random.RandomState(1)
    Image.fromarray((rs.rand(100, 140, 3) * 255).astype(np.uint8)).save(base / "images" / "a.png")
    Image.fromarray((rs.rand(100, 140) * 255).astype(np.uint8)).save(base / "disps" / "a.png")
    m = np.zeros((100, 140), np.uint8); m[30:60, 40:90] = 1
    Image.fromarray(m).save(base / "masks" / "a.png")
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(root, "gen_3dphoto_dynamic.py"), "--base", str(base), "--out", str(out), "--width", "128",
                        "--height", "128", "--repeat", "1", "--planes", "8", "--inpaint", "hip", "--mpi-from", "model", "--ckpt_path", "random:3",
                        "--model-engine", "torch"],                                        # the opt-out: every convolution on torch / MIOpen
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    flow = io_formats.read_flo(str(out / "flows" / "a_0.flo"))
    assert flow.shape == (128, 128, 2) and np.isfinite(flow).all()
    assert Image.open(out / "dst_images" / "a_0.png").size == (128, 128)


def test_cli_with_network_on_hip_engine(dev, tmp_path):
    """The default producer (--model-engine hip): the per-plane networks on the MFMA engine, replayed from one hipGraph per image; two
    images so the captured graph is replayed with new inputs (replay == eager is checked in tests/test_conv_engine.py)."""
    import subprocess, sys, os
    from PIL import Image
    from mpiflow_amd import io_formats
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = tmp_path / "data"
    for d in ("images", "disps", "masks"):
        (base / d).mkdir(parents=True)
    rs = np.random.RandomState(2)
    for n in ("a", "b"):
        Image.fromarray((rs.rand(100, 140, 3) * 255).astype(np.uint8)).save(base / "images" / (n + ".png"))
        Image.fromarray((rs.rand(100, 140) * 255).astype(np.uint8)).save(base / "disps" / (n + ".png"))
        m = np.zeros((100, 140), np.uint8); m[30:60, 40:90] = 1
        Image.fromarray(m).save(base / "masks" / (n + ".png"))
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(root, "gen_3dphoto_dynamic.py"), "--base", str(base), "--out", str(out), "--width", "128",
                        "--height", "128", "--repeat", "1", "--planes", "8", "--inpaint", "hip", "--mpi-from", "model", "--ckpt_path", "random:3"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    flows = [io_formats.read_flo(str(out / "flows" / (n + "_0.flo"))) for n in ("a", "b")]
    for f in flows:
        assert f.shape == (128, 128, 2) and np.isfinite(f).all()
    assert not np.array_equal(flows[0], flows[1])                         # the replay really saw the second image


def test_cli_precise_engine_matches_the_torch_fp32_producer(dev, tmp_path):
    """gen_3dphoto_dynamic.py --model-engine hip --model-dtype fp32 | fp32-mfma | fp64: the parity-grade producer (every convolution on mpf_pconv) behind the
    reference's entry point.  Same image, same seed, same poses as --model-engine torch (the fp32 torch modules): the written flows agree to what two
    fp32 evaluations of the network allow (tests/test_precise_engine.py), the fp32 and fp64 engines agree with each other more closely still, and
    bf16 (a torch autocast dtype) is refused for the hip engine."""
    import subprocess, sys, os
    from PIL import Image
    from mpiflow_amd import io_formats
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = tmp_path / "data"
    for d in ("images", "disps", "masks"):
        (base / d).mkdir(parents=True)
    rs = np.random.RandomState(4)
    yy, xx = np.mgrid[0:100, 0:140]
    img = np.clip(0.5 + 0.3 * np.sin(xx / 9.0) + 0.2 * np.cos(yy / 7.0), 0, 1)
    Image.fromarray((np.stack([img, np.roll(img, 5, 1), np.roll(img, 9, 0)], -1) * 255).astype(np.uint8)).save(base / "images" / "a.png")
    Image.fromarray((255 * (0.1 + 0.8 * yy / 100)).astype(np.uint8)).save(base / "disps" / "a.png")
    m = np.zeros((100, 140), np.uint8); m[30:60, 40:90] = 1
    Image.fromarray(m).save(base / "masks" / "a.png")
    flows = {}
    common = [sys.executable, os.path.join(root, "gen_3dphoto_dynamic.py"), "--base", str(base), "--width", "128", "--height", "128", "--repeat", "2", "--planes", "8",
              "--inpaint", "none", "--mpi-from", "model", "--ckpt_path", "random:3"]
    for name, extra in (("torch", ["--model-engine", "torch"]), ("fp32", ["--model-engine", "hip", "--model-dtype", "fp32"]),
                        ("fp32-mfma", ["--model-engine", "hip", "--model-dtype", "fp32-mfma"]), ("fp64", ["--model-engine", "hip", "--model-dtype", "fp64"])):
        out = tmp_path / ("out_" + name)
        r = subprocess.run(common + ["--out", str(out)] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        flows[name] = np.stack([io_formats.read_flo(str(out / "flows" / ("a_%d.flo" % k))) for k in range(2)])
        assert flows[name].shape == (2, 128, 128, 2) and np.isfinite(flows[name]).all()
        assert Image.open(out / "dst_images" / "a_1.png").size == (128, 128)
    scale = float(np.abs(flows["torch"]).max())
    d32, d64, dd = (float(np.abs(flows[a] - flows[b]).mean()) for a, b in (("fp32", "torch"), ("fp64", "torch"), ("fp32", "fp64")))
    assert scale > 0.5 and d32 < 2e-4 * max(scale, 1.0) and d64 < 2e-4 * max(scale, 1.0) and dd <= max(d32, d64), (scale, d32, d64, dd)
    dm = float(np.abs(flows["fp32-mfma"] - flows["fp64"]).mean())                   # the two fp32 forms (bf16-piece products / fp32 products) are different kernels
    assert dm <= max(d32, d64) and not np.array_equal(flows["fp32-mfma"], flows["fp32"]), (dm, d32, d64)
    r = subprocess.run(common + ["--out", str(tmp_path / "x"), "--model-engine", "hip", "--model-dtype", "bf16"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "bf16" in (r.stderr + r.stdout)


def test_hard_flow_entry_point(dev, oracle):
    """hard_flow=True: flow of the arg-max-weight plane (mpi_rendering.py:126-130).  A 1-ulp exp difference (the reference's exp is MKL's)
    can move the arg-max between two planes whose weights are equal to within rounding; every pixel that differs from the reference's
    golden must be such a PROVEN near-tie (top-2 source-frame weights within 2e-6 relative), and there may be only a handful."""
    from mpiflow_amd import synth
    from mpiflow_amd.utils import utils as U
    g = load_golden("hard_flow")
    S, H, W = int(g["S"]), int(g["H"]), int(g["W"])
    inp = synth.make_inputs(S, H, W, seed=int(g["seed"]), kind="white")

    class Opt:
        ext_cz = 0.15

    random.seed(int(g["pose_seed"]))
    K = T(inp["K"], dev)[None]
    flow_mix, _, _, _ = U.render_3dphoto_dynamic(Opt, T(inp["image"], dev)[None], T(inp["obj_mask"], dev)[None, None], None, T(inp["mpi"], dev)[None],
                                                 T(inp["disparity"], dev)[None], K, K, name="g.png", hard_flow=True, inpaint="none")
    err = np.abs(flow_mix - g["flow_mix"]).max(axis=-1)
    bad = err >= 1e-4
    d = oracle.plane_depths(inp["disparity"])
    w = oracle.volume_render(None, inp["mpi"][:, 3:], oracle.src_xyz(oracle.k_inverse(inp["K"]), d, H, W))["weights"]      # [S,H,W]
    top2 = np.sort(w, axis=0)[-2:]
    near_tie = (top2[1] - top2[0]) <= 2e-6 * np.maximum(top2[1], 1e-30)
    assert (bad & ~near_tie).sum() == 0, "hard flow differs at %d pixels that are not arg-max ties" % int((bad & ~near_tie).sum())
    assert bad.sum() <= max(4, int(0.005 * H * W)), "hard flow differs at %d of %d pixels" % (int(bad.sum()), H * W)
    print("\n[hard_flow] %d of %d pixels pick another plane than the reference; all are arg-max near-ties (%d near-tie pixels in the frame)"
          % (int(bad.sum()), H * W, int(near_tie.sum())))


@pytest.mark.parametrize("S,H,W", [(8, 32, 48), (20, 23, 37), (1, 9, 11), (64, 40, 72), (5, 17, 300)])
def test_fused_hard_flow_equals_the_materialised_form(dev, S, H, W):
    """mpf_src_flow_hard (one pass over the sigma planes, nothing per-plane materialised) == the reference-shaped path (mpf_homography_flow +
    mpf_volume_render(hard): per-plane flows written and re-read), bit for bit, for one, two and three poses, on the [S,4,H,W] stack and on a bare
    sigma tensor; ties take the first maximal plane (a stack of equal sigmas: plane 0 wins wherever its weight is the largest)."""
    from mpiflow_amd import host_math, ops, pipeline, synth
    inp = synth.make_inputs(S, H, W, seed=S * 7 + W, kind="white")
    rng = random.Random(S + H)
    poses = [host_math.generate_random_pose(0.15, rng=rng), host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng), host_math.generate_random_pose(0.3, rng=rng)]
    mpi = T(inp["mpi"], dev)
    for n in (1, 2, 3):
        got = pipeline.hard_flows(mpi, inp["disparity"], inp["K"], poses[:n])
        want = pipeline.hard_flows(mpi, inp["disparity"], inp["K"], poses[:n], generic=True)
        assert tuple(got.shape) == (n, 2, H, W) and torch.equal(got.view(torch.int32), want.view(torch.int32)), n
    k_inv, d = host_math.k_inverse(inp["K"]), host_math.plane_depths(inp["disparity"])
    H_ts, _ = host_math.homographies(poses[0], k_inv, inp["K"], d)
    bare = ops.src_flow_hard(mpi[:, 3:4].contiguous(), k_inv, d, H_ts.unsqueeze(0))
    assert torch.equal(bare, pipeline.hard_flows(mpi, inp["disparity"], inp["K"], poses[:1]))
    flat = mpi.clone()
    flat[:, 3] = 0.5                                                     # equal sigmas: weights fall monotonically after the first plane that absorbs anything
    a, b = pipeline.hard_flows(flat, inp["disparity"], inp["K"], poses[:2]), pipeline.hard_flows(flat, inp["disparity"], inp["K"], poses[:2], generic=True)
    assert torch.equal(a, b)


def test_pipeline_is_deterministic_and_graph_capturable(dev):
    """Two eager runs are bit-identical, and the two fused launches replay from a captured HIP graph with the same result
    (the C ABI promises: asynchronous on the given stream, no allocation, no synchronisation)."""
    from mpiflow_amd import host_math, ops, pipeline, synth
    S, H, W = 16, 48, 80
    inp = synth.make_inputs(S, H, W, seed=21)
    mpi, img, om = T(inp["mpi"], dev), T(inp["image"], dev), T(inp["obj_mask"], dev)
    r = pipeline.PairRenderer(S, H, W, dev)
    rng = random.Random(5)
    Gd, Gc = host_math.generate_random_pose(0.15, rng=rng), host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)
    prep = r.prepare(inp["K"], inp["disparity"], [Gc, Gd])
    r.run(mpi, img, prep, om)
    torch.cuda.synchronize()
    first = [r.views[0]["rgb"].clone(), r.views[1]["objmask"].clone(), r.flows.clone(), r.src_u8.clone(), r.views[1]["rgb_u8"].clone()]
    r.run(mpi, img, prep, om)
    torch.cuda.synchronize()
    for a, b in zip(first, [r.views[0]["rgb"], r.views[1]["objmask"], r.flows, r.src_u8, r.views[1]["rgb_u8"]]):
        assert torch.equal(a, b)
    for v in r.views:
        v["rgb"].zero_(); v["objmask"].zero_()
    r.flows.zero_()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            r.run(mpi, img, prep, om)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(first, [r.views[0]["rgb"], r.views[1]["objmask"], r.flows, r.src_u8, r.views[1]["rgb_u8"]]):
        assert torch.equal(a, b)


def test_full_size_properties_c5(dev):
    """BASELINE config 5 shape (128 x 1024 x 1536): identity pose -> zero flow, all planes valid, first blended plane == image."""
    from mpiflow_amd import host_math, ops, synth
    S, H, W = 128, 1024, 1536
    g = torch.Generator(device=dev).manual_seed(0)
    mpi = torch.rand((S, 4, H, W), generator=g, device=dev)
    mpi[:, 3] = (torch.relu(3 * torch.randn((S, 1, 1), generator=g, device=dev) - 3) * 0.02 + 1e-4).expand(S, H, W)
    img = torch.rand((3, H, W), generator=g, device=dev)
    K = synth.intrinsics(H, W)
    k_inv = host_math.k_inverse(K)
    d = host_math.plane_depths(synth.plane_disparities(S))
    G = torch.eye(4)
    H_ts, H_st = host_math.homographies(G, k_inv, K, d)
    rgba = ops.alloc_rgba_stack(S, H, W, dev)
    a = ops.src_blend_flow(mpi, img, k_inv, d, H_ts[None], out_rgba=rgba)
    assert float(a["flows"].abs().max()) < 2e-3
    assert torch.equal(rgba[0, :, :, :3].permute(2, 0, 1), img)
    v = ops.warp_composite(rgba, None, H_st, k_inv, G, d, interleaved=2)
    assert float(v["tgt_mask"].min()) == S and float(v["tgt_mask"].max()) == S
    assert bool(torch.isfinite(v["rgb"]).all()) and float(v["rgb"].min()) >= 0 and float(v["rgb"].max()) <= 1.0 + 1e-5


def test_merge_depth_ordered_kernel_vs_restatement(dev):
    """mpf_merge_depth_ordered against the numpy restatement of "utils/utils copy.py":278-303 on adversarial inputs: masks at 0, at the
    threshold and NaN, equal depths, NaN depths."""
    from mpiflow_amd import ops
    from oracle import mpi_oracle as o
    rs = np.random.RandomState(5)
    H, W = 37, 53
    frame, frame_dyn = rs.rand(3, H, W).astype(np.float32) * 1.2 - 0.1, rs.rand(3, H, W).astype(np.float32)
    vals = np.array([0.0, -0.0, 1e-30, 0.5, np.float32(0.99), np.nextafter(np.float32(0.99), np.float32(0)), 1.0, np.nan], np.float32)
    mask, mask_dyn = vals[rs.randint(0, len(vals), (H, W))], vals[rs.randint(0, len(vals), (H, W))]
    depth = rs.rand(H, W).astype(np.float32) * 3
    depth_dyn = np.where(rs.rand(H, W) < 0.3, depth, rs.rand(H, W).astype(np.float32) * 3).astype(np.float32)
    depth[rs.rand(H, W) < 0.05] = np.nan
    want, want_mask = o.merge_depth_ordered(frame, frame_dyn, mask, mask_dyn, depth, depth_dyn)
    got, got_mask = ops.merge_depth_ordered(T(frame, dev), T(frame_dyn, dev), T(mask, dev), T(mask_dyn, dev), T(depth, dev), T(depth_dyn, dev),
                                            want_depth_mask=True)
    assert bits_equal(N(got), want) == 0 and bits_equal(N(got_mask).astype(bool), want_mask) == 0
    assert want_mask.sum() > 100
    assert bits_equal(N(ops.merge_depth_ordered(T(frame, dev), T(frame_dyn, dev), T(mask, dev), T(mask_dyn, dev), T(depth, dev), T(depth_dyn, dev))), want) == 0


@pytest.mark.parametrize("case", [0, 1])
def test_older_module_entry_point_with_depth_ordered_frame(dev, case):
    """utils_copy.render_3dphoto_dynamic = the reference's "utils/utils copy.py":164-326 (own pose constants, depth-ordered frame),
    against what the reference produced under the same random.seed: merged frame / flow as in the v2 test; the depth-ordered frame
    byte-exact wherever the depth comparison is not a float near-tie and the masks are off the threshold margin."""
    from mpiflow_amd.utils import utils_copy as UC
    g = load_golden("copy_variant")
    p = "c%d_" % case
    random.seed(int(g[p + "pose_seed"]))
    K = T(g[p + "K"], dev)[None]
    flow_mix, src_np, inpainted, res = UC.render_3dphoto_dynamic(
        T(g[p + "image"], dev)[None], T(g[p + "obj_mask"], dev)[None, None], None, T(g[p + "mpi"], dev)[None], T(g[p + "disparity"], dev)[None],
        K, K, data_path="outputs", name="demo.png", inpaint="none")
    assert bits_equal(src_np, g[p + "src_np"]) == 0
    th = np.float32(0.99)
    margin = (np.abs(g[p + "cam_objmask"].astype(np.float64) - th) < 1e-5) | (np.abs(g[p + "dyn_objmask"].astype(np.float64) - th) < 1e-5)
    ok = ~margin
    assert max_abs(flow_mix[ok], g[p + "flow_mix"][ok]) < 1e-4
    assert np.abs(inpainted[ok].astype(np.int32) - g[p + "frame_mix"][ok].astype(np.int32)).max() <= 1          # inpaint="none": the merged frame
    zc, zd = g[p + "cam_depth"], g[p + "dyn_depth"]
    assert max_abs(res["depth"], zc) < 2e-5 * max(1.0, float(np.abs(zc).max())) and max_abs(res["depth_dync"], zd) < 2e-5 * max(1.0, float(np.abs(zd).max()))
    tie = np.abs(zc - zd) < 1e-4 * np.maximum(np.abs(zc), 1.0)
    # a mask that is zero in the reference must be zero here too for the non-zero test to agree: the composited masks are sums of
    # non-negative terms, exactly 0 iff every term is
    sure = ok & ~tie
    want_pick = (zc > zd) & (g[p + "cam_objmask"] != 0) & (g[p + "dyn_objmask"] != 0)
    assert (res["depth_mask"][sure] == want_pick[sure]).all()
    d = np.abs(res["frame_mix_depth"][sure].astype(np.int32) - g[p + "frame_mix_depth"][sure].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    changed = (g[p + "frame_mix_depth"] != g[p + "frame_mix"]).any(-1)
    assert (changed & sure).sum() >= 5 and (res["frame_mix_depth"][changed & sure] != inpainted[changed & sure]).any(-1).sum() >= 5
    assert res["frame_mix_depth_inpainted"].shape == g[p + "frame_mix"].shape
