"""Output side of a pair: PNG writer, hole fill, streaming ring, input prefetch."""
import io
import os

import numpy as np
import pytest
import torch

from conftest import bits_equal, load_golden


def _rgb(h, w, seed=0):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 0.5 + 0.3 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
    return (np.clip(base[..., None] + 0.1 * rs.randn(h, w, 3), 0, 1) * 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(1, 1), (5, 3), (48, 64), (37, 1280)])
def test_png_writer_round_trips_through_pillow(tmp_path, shape):
    from PIL import Image
    from mpiflow_amd import io_formats
    bgr = _rgb(*shape)
    path = str(tmp_path / "f.png")
    io_formats.write_png_bgr(path, bgr)
    back = np.array(Image.open(path))
    assert back.shape == bgr.shape and np.array_equal(back, bgr[:, :, ::-1])          # file stores RGB (cv2.imwrite semantics)
    data = io_formats.png_from_scanlines(io_formats.filter_up_rgb(bgr))
    assert np.array_equal(np.array(Image.open(io.BytesIO(data))), bgr)
    assert data[:8] == b"\x89PNG\r\n\x1a\n"


def test_input_prefetcher_yields_what_the_reference_loaders_decode(tmp_path):
    """u8 arrays of the OWNED images, in listing order: image = PIL RGB (utils/utils.py:36), disparity = cv2.imread(path, 0)
    (:43; here 8-bit grey files, where that is the stored bytes), mask = PIL "L" (gen_3dphoto_dynamic_v2.py:83)."""
    from PIL import Image
    from mpiflow_amd import io_formats
    from mpiflow_amd.utils import utils as U
    dirs = {}
    for d in ("images", "disps", "masks"):
        dirs[d] = tmp_path / d
        dirs[d].mkdir()
    names = ["b.png", "a.png", "c.png"]
    for k, n in enumerate(names):
        Image.fromarray(_rgb(20, 30, k)).save(dirs["images"] / n)
        Image.fromarray(_rgb(20, 30, 10 + k)[..., 0]).save(dirs["disps"] / n)
        Image.fromarray((_rgb(20, 30, 20 + k)[..., 0] // 100).astype(np.uint8)).save(dirs["masks"] / n)
    order = sorted(names)
    got = list(io_formats.InputPrefetcher(order, str(dirs["images"]), str(dirs["disps"]), str(dirs["masks"]), [0, 2], pin=False))
    assert [g["i"] for g in got] == [0, 2] and [g["name"] for g in got] == [order[0], order[2]]
    for g in got:
        n = g["name"]
        assert g["error"] is None
        assert np.array_equal(g["ids_u8"], np.array(Image.open(dirs["masks"] / n).convert("L")))
        assert np.array_equal(g["rgb_u8"], np.array(Image.open(dirs["images"] / n).convert("RGB")))
        assert np.array_equal(g["disp_u8"], np.array(Image.open(dirs["disps"] / n)))
        # the float tensors of the reference's loaders are these bytes / 255
        assert np.array_equal(U.image_to_tensor(str(dirs["images"] / n))[0].numpy(), g["rgb_u8"].transpose(2, 0, 1).astype(np.float32) / np.float32(255))
        assert np.array_equal(U.disparity_to_tensor(str(dirs["disps"] / n))[0, 0].numpy(), (g["disp_u8"] / 255).astype(np.float32))


def test_input_prefetcher_accepts_files_of_different_sizes(tmp_path):
    """The reference resizes image, disparity and mask INDEPENDENTLY to (height, width) (gen_3dphoto_dynamic_v2.py:82-89, :104-105), so
    a disparity saved at another resolution (MiDaS / DPT network size) is valid input - not an error, not a skipped image."""
    from PIL import Image
    from mpiflow_amd import io_formats
    for d in ("images", "disps", "masks"):
        (tmp_path / d).mkdir()
    Image.fromarray(_rgb(20, 30, 1)).save(tmp_path / "images" / "x.png")
    Image.fromarray(_rgb(12, 16, 2)[..., 0]).save(tmp_path / "disps" / "x.png")
    Image.fromarray((_rgb(24, 36, 3)[..., 0] // 100).astype(np.uint8)).save(tmp_path / "masks" / "x.png")
    (g,) = list(io_formats.InputPrefetcher(["x.png"], str(tmp_path / "images"), str(tmp_path / "disps"), str(tmp_path / "masks"), [0], pin=False))
    assert g["error"] is None
    assert g["rgb_u8"].shape == (20, 30, 3) and g["disp_u8"].shape == (12, 16) and g["ids_u8"].shape == (24, 36)


def test_input_prefetcher_reports_errors_per_image(tmp_path):
    from PIL import Image
    from mpiflow_amd import io_formats
    for d in ("images", "disps", "masks"):
        (tmp_path / d).mkdir()
        Image.fromarray(_rgb(8, 9, 1)[..., 0]).save(tmp_path / d / "ok.png")
    Image.fromarray(_rgb(8, 9, 1)).save(tmp_path / "images" / "ok.png")
    (tmp_path / "masks" / "bad.png").write_bytes(b"not a png")
    got = list(io_formats.InputPrefetcher(["bad.png", "missing.png", "ok.png"], str(tmp_path / "images"), str(tmp_path / "disps"),
                                          str(tmp_path / "masks"), [0, 1, 2], pin=False))
    assert got[0]["error"] is not None and got[1]["error"] is not None and got[2]["error"] is None
    assert io_formats.mask_max_of_file(str(tmp_path / "masks" / "bad.png")) == -1


@pytest.mark.parametrize("job", ["cpu_job", pytest.param("gpu_box_job", marks=pytest.mark.gpu)])
def test_disparity_decode_follows_cv2_imread_grayscale(tmp_path, job):
    """cv2.imread(path, 0) (utils/utils.py:43): 16-bit grey -> high byte (PIL's convert("L") would saturate), colour PNG -> libpng's
    fixed-point grey, 8-bit grey -> stored bytes.  Compared with the real cv2 when it is installed - in the CPU job and again on the GPU
    box, a second machine that may carry OpenCV (needs no GPU)."""
    from PIL import Image
    from mpiflow_amd import io_formats
    ramp = (np.arange(300 * 200, dtype=np.uint32).reshape(200, 300) * 65535 // (300 * 200 - 1)).astype(np.uint16)
    Image.fromarray(ramp).save(tmp_path / "d16.png")
    got = io_formats.read_disparity_u8(str(tmp_path / "d16.png"))
    assert got.dtype == np.uint8 and np.array_equal(got, (ramp >> 8).astype(np.uint8))
    assert (got == 255).mean() < 0.01                      # not saturated: the ADVICE.md failure mode (99.6 % of pixels at 255)
    g8 = _rgb(40, 50, 3)[..., 0]
    Image.fromarray(g8).save(tmp_path / "d8.png")
    assert np.array_equal(io_formats.read_disparity_u8(str(tmp_path / "d8.png")), g8)
    rgb = _rgb(40, 50, 4)
    Image.fromarray(rgb).save(tmp_path / "dc.png")
    c = rgb.astype(np.uint32)
    assert np.array_equal(io_formats.read_disparity_u8(str(tmp_path / "dc.png")),
                          ((c[..., 0] * 9798 + c[..., 1] * 19235 + c[..., 2] * 3735 + 16384) >> 15).astype(np.uint8))
    try:
        import cv2
    except Exception:
        return                                             # OpenCV absent: the decoder's parity with cv2 stays unpinned
    for f in ("d16.png", "d8.png", "dc.png"):
        assert np.array_equal(io_formats.read_disparity_u8(str(tmp_path / f)), cv2.imread(str(tmp_path / f), 0)), f


def test_input_stage_oracle_matches_the_reference_golden(oracle):
    """ToTensor / `/255` / (ids == k) + F.interpolate(bilinear, align_corners=True): the oracle's restatement vs what the
    reference's own loaders and torch-CPU produced (tests/golden/make_golden.py: inputs), bit for bit, for both ATen kernels."""
    g = load_golden("input_stage")
    for tag, size in (("big", (96, 160)), ("small", (40, 72))):
        out = oracle.prepare_inputs(g["rgb_u8"], g["disp_u8"], size=size)
        assert bits_equal(out["image"], g["image_" + tag]) == 0 and bits_equal(out["disp"], g["disp_" + tag]) == 0
        for k in (1, 2, 3):
            m = oracle.prepare_inputs(ids_u8_hw=g["ids_u8"], obj_index=k, size=size)["mask"]
            assert bits_equal(m, g["mask%d_%s" % (k, tag)]) == 0
    assert int(g["mask_max"]) == int(g["ids_u8"].max())


@pytest.mark.gpu
def test_input_stage_kernel_matches_the_reference_golden(oracle):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mpiflow_amd import ops
    dev = torch.device("cuda:0")
    g = load_golden("input_stage")
    rgb, dsp, ids = (torch.from_numpy(g[k]).to(dev) for k in ("rgb_u8", "disp_u8", "ids_u8"))
    for tag, size in (("big", (96, 160)), ("small", (40, 72))):
        out = ops.prepare_inputs(rgb_u8=rgb, disp_u8=dsp, size=size)
        assert bits_equal(out["image"].cpu().numpy(), g["image_" + tag]) == 0
        assert bits_equal(out["disp"].cpu().numpy(), g["disp_" + tag]) == 0
        for k in (1, 2, 3):
            m = ops.prepare_inputs(ids_u8=ids, obj_index=k, size=size)["mask"]
            assert bits_equal(m.cpu().numpy(), g["mask%d_%s" % (k, tag)]) == 0
    # a KITTI-sized frame against the oracle (itself pinned above and against torch in the build container)
    rs = np.random.RandomState(7)
    rgb = (rs.rand(375, 1242, 3) * 256).astype(np.uint8)
    dsp = (rs.rand(375, 1242) * 256).astype(np.uint8)
    ids = (rs.rand(375, 1242) * 4).astype(np.uint8)
    want = oracle.prepare_inputs(rgb, dsp, ids, obj_index=2, size=(384, 1280))
    got = ops.prepare_inputs(torch.from_numpy(rgb).to(dev), torch.from_numpy(dsp).to(dev), torch.from_numpy(ids).to(dev), obj_index=2, size=(384, 1280))
    for k in ("image", "disp", "mask"):
        assert bits_equal(got[k].cpu().numpy(), want[k]) == 0, k
def _peel_reference(img, hole):
    """Plain restatement of the onion peel: repeat full-image passes on a copy of the previous state."""
    img, hole = img.astype(np.int64).copy(), hole.astype(bool).copy()
    H, W = hole.shape
    while True:
        new_img, new_hole, progress = img.copy(), hole.copy(), False
        for y, x in zip(*np.nonzero(hole)):
            acc, cnt = np.zeros(3, np.int64), 0
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    yy, xx = y + dy, x + dx
                    if (dy or dx) and 0 <= yy < H and 0 <= xx < W and not hole[yy, xx]:
                        acc += img[yy, xx]
                        cnt += 1
            if cnt:
                new_img[y, x] = (acc + cnt // 2) // cnt
                new_hole[y, x] = False
                progress = True
        img, hole = new_img, new_hole
        if not progress:
            return img.astype(np.uint8), hole.astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["blobs", "all_hole", "no_hole", "border"])
def test_hole_fill_matches_the_peel_definition(case):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mpiflow_amd import ops
    dev = torch.device("cuda:0")
    H, W = 41, 67
    img = _rgb(H, W, 3)
    rs = np.random.RandomState(5)
    hole = np.zeros((H, W), np.uint8)
    if case == "blobs":
        hole[5:20, 8:30] = 1
        hole[25:40, 40:66] = 1
        hole[rs.rand(H, W) < 0.05] = 1
    elif case == "all_hole":
        hole[:] = 1
    elif case == "border":
        hole[:, :9] = 1
        hole[-6:, :] = 1
    ref_img, ref_hole = _peel_reference(img, hole)
    hole_out = torch.empty((H, W), dtype=torch.uint8, device=dev)
    got = ops.fill_holes(torch.from_numpy(img).to(dev), torch.from_numpy(hole).to(dev), hole_out=hole_out)
    assert np.array_equal(hole_out.cpu().numpy(), ref_hole)
    known = ref_hole == 0
    assert np.array_equal(got.cpu().numpy()[known], ref_img[known])


@pytest.mark.gpu
def test_png_scanlines_on_device_and_output_ring(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    from mpiflow_amd import io_formats, ops
    dev = torch.device("cuda:0")
    H, W = 48, 80
    ring = io_formats.OutputRing(H, W, dev, slots=3, threads=2)
    frames, flows = [], []
    for k in range(7):                                   # more pairs than slots: exercises the back-pressure path
        bgr = _rgb(H, W, k)
        flow = np.random.RandomState(k).randn(H, W, 2).astype(np.float32)
        scan = ops.png_scanlines(torch.from_numpy(bgr).to(dev))
        assert np.array_equal(scan.cpu().numpy(), io_formats.filter_up_rgb(bgr[:, :, ::-1]))
        ring.submit_pair(torch.from_numpy(flow).to(dev), scan, str(tmp_path / ("f%d.flo" % k)), str(tmp_path / ("d%d.png" % k)))
        frames.append(bgr)
        flows.append(flow)
    ring.submit_source(ops.png_scanlines(torch.from_numpy(frames[0]).to(dev)), [str(tmp_path / ("s%d.png" % r)) for r in range(3)])
    ring.close()
    for k in range(7):
        assert np.array_equal(np.array(Image.open(tmp_path / ("d%d.png" % k)))[:, :, ::-1], frames[k])
        assert np.array_equal(io_formats.read_flo(str(tmp_path / ("f%d.flo" % k))), flows[k])
    for r in range(3):
        assert np.array_equal(np.array(Image.open(tmp_path / ("s%d.png" % r)))[:, :, ::-1], frames[0])
    assert sorted(os.listdir(tmp_path)) == sorted(["f%d.flo" % k for k in range(7)] + ["d%d.png" % k for k in range(7)] + ["s%d.png" % r for r in range(3)])


@pytest.mark.gpu
def test_output_ring_takes_a_pair_slab_in_one_copy(tmp_path):
    """submit_pair_fill(slab=...): flow | frame | hole mask leave the GPU as ONE device-to-host copy of the ops.pair_slab they are views of - same files
    as the three separate copies."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    from mpiflow_amd import io_formats, ops
    dev = torch.device("cuda:0")
    H, W = 40, 72
    seen = []

    def fill(frame, hole):                                # the writer thread's hook sees exactly what was rendered
        seen.append((frame.copy(), hole.copy()))
        out = frame.copy()
        out[hole > 0] = 7
        return out
    ring = io_formats.OutputRing(H, W, dev, slots=2, threads=1, host_fill=fill)
    want = []
    for k in range(5):
        bgr, flow = _rgb(H, W, k), np.random.RandomState(k).randn(H, W, 2).astype(np.float32)
        hole = (np.random.RandomState(10 + k).rand(H, W) < 0.1).astype(np.uint8)
        slab, (f, fr, ho) = ops.pair_slab(H, W, dev)
        f.copy_(torch.from_numpy(flow)); fr.copy_(torch.from_numpy(bgr)); ho.copy_(torch.from_numpy(hole))
        if k % 2:
            ring.submit_pair_fill(f, fr, ho, str(tmp_path / ("f%d.flo" % k)), str(tmp_path / ("d%d.png" % k)), slab=slab)
        else:
            ring.submit_pair_fill(f, fr, ho, str(tmp_path / ("f%d.flo" % k)), str(tmp_path / ("d%d.png" % k)))
        want.append((bgr, flow, hole))
    ring.close()
    assert len(seen) == 5
    for k, (bgr, flow, hole) in enumerate(want):
        assert np.array_equal(io_formats.read_flo(str(tmp_path / ("f%d.flo" % k))), flow)
        filled = bgr.copy()
        filled[hole > 0] = 7
        assert np.array_equal(np.array(Image.open(tmp_path / ("d%d.png" % k)))[:, :, ::-1], filled)


@pytest.mark.gpu
def test_pair_stats_kernel(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mpiflow_amd import pipeline
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(8)
    st = pipeline.DeviceStats(dev)
    tot = dict(sum=0.0, hole=0.0, mx=0.0, neg=-np.inf)
    for k in range(3):
        flow = (rs.randn(37, 53, 2) * (5 + 10 * k)).astype(np.float32)
        fill = (rs.rand(37, 53) < 0.1).astype(np.uint8)
        st.add(torch.from_numpy(flow).to(dev), torch.from_numpy(fill).to(dev))
        mag = np.sqrt((flow.astype(np.float64) ** 2).sum(-1))
        tot["sum"] += mag.sum(); tot["hole"] += fill.sum(); tot["mx"] = max(tot["mx"], mag.max()); tot["neg"] = max(tot["neg"], (-flow).max())
    r = st.result(3)
    assert r["pairs"] == 3 and r["hole_px"] == tot["hole"]
    assert abs(r["sum_flow_mag"] - tot["sum"]) < 1e-3 * tot["sum"] * 1e-3
    assert abs(r["max_flow_mag"] - tot["mx"]) < 1e-4 and r["neg_min_flow"] == float(tot["neg"])


@pytest.mark.gpu
def test_stream_probe_copies_and_validates():
    """mpf_stream_probe (bench.py's on-box HBM reference): mode 1 is an exact copy, mode 0 leaves its sink alone, bad arguments
    are refused with an error code."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes
    from mpiflow_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    a = torch.randn(1 << 20, device=dev)
    b = torch.zeros_like(a)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    assert lib.mpf_stream_probe(p(a), p(b), a.numel() * 4, 1, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    sink = torch.full((4,), 7.0, device=dev)
    assert lib.mpf_stream_probe(p(a), p(sink), a.numel() * 4, 0, st) == 0
    torch.cuda.synchronize()
    assert (sink == 7.0).all()
    assert lib.mpf_stream_probe(p(a), p(b), 24, 1, st) != 0          # not a multiple of 16
    assert lib.mpf_stream_probe(p(a), p(b), 64, 2, st) != 0          # unknown mode
