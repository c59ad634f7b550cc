"""CPU-only tests of the host logic: small-matrix maths vs goldens from the reference, parameter packing, the pose
schedule replay, image sharding and the world_size-2 statistics all-reduce (gloo)."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import bits_equal, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["tiny_white", "tiny_smooth"])
def test_host_matrices_match_reference(name):
    from mpiflow_amd import host_math
    g = load_golden(name)
    k_inv = host_math.k_inverse(g["K"])
    assert bits_equal(k_inv.numpy(), g["k_inv"]) == 0
    d = host_math.plane_depths(g["disparity"])
    assert bits_equal(d.numpy(), g["depth_S"]) == 0
    H_ts, H_st = host_math.homographies(g["G_cam"], k_inv, g["K"], d)
    assert bits_equal(H_ts.numpy(), g["H_tgt_src_cam"]) == 0
    assert bits_equal(H_st.numpy(), g["H_src_tgt_cam"]) == 0


def test_pose_profiles_of_the_other_reference_variants():
    """utils/utils_coco.py and "utils/utils copy.py" carry their own sampler constants: same RNG stream,
    bit-identical 4x4 poses for the goldens recorded from those modules."""
    import random as _random
    from mpiflow_amd import host_math
    g = load_golden("pose_schedule_variants")
    for profile in ("coco", "copy"):
        rng = _random.Random(int(g["seed"]))
        for k in range(g["G_dyn_" + profile].shape[0]):
            dyn = host_math.generate_random_pose(rng=rng, profile=profile)
            cam = host_math.generate_random_pose(base_motions=[0, 0, 0], rng=rng, profile=profile)
            assert np.array_equal(dyn.numpy(), g["G_dyn_" + profile][k]), (profile, k)
            assert np.array_equal(cam.numpy(), g["G_cam_" + profile][k]), (profile, k)


def test_pose_schedule_matches_reference_rng_stream():
    from mpiflow_amd import pipeline
    g = load_golden("pose_schedule")
    sched = pipeline.pose_schedule(int(g["seed"]), g["G_dyn"].shape[0], float(g["ext_cz"]))
    for i, (cam, dyn) in enumerate(sched):
        assert bits_equal(dyn.numpy(), g["G_dyn"][i]) == 0
        assert bits_equal(cam.numpy(), g["G_cam"][i]) == 0
    # and through the module-level `random`, as the reference's entry point uses it
    from mpiflow_amd.utils import utils as U
    random.seed(int(g["seed"]))
    assert bits_equal(U.generate_random_pose(float(g["ext_cz"])).numpy(), g["G_dyn"][0]) == 0
    assert bits_equal(U.generate_random_pose(float(g["ext_cz"]), base_motions=[0, 0, 0]).numpy(), g["G_cam"][0]) == 0


def test_geometry_pose_algebra_matches_reference():
    from mpiflow_amd import geometry
    g = load_golden("geometry")
    M = geometry.transformation_from_parameters(torch.from_numpy(g["axisangle"]), torch.from_numpy(g["translation"]))
    Mi = geometry.transformation_from_parameters(torch.from_numpy(g["axisangle"]), torch.from_numpy(g["translation"]), invert=True)
    assert bits_equal(M.numpy(), g["M"]) == 0 and bits_equal(Mi.numpy(), g["M_inv"]) == 0


def test_inverse_raises_like_the_reference():
    from mpiflow_amd.utils.mpi.homography_sampler import inverse
    with pytest.raises(Exception, match="Matrix inverse contains nan!"):
        inverse(torch.full((2, 3, 3), float("nan"), dtype=torch.float64))


def test_param_packing_layout():
    from mpiflow_amd import host_math, ops
    S = 5
    k_inv = torch.arange(9, dtype=torch.float32).reshape(3, 3)
    G = torch.arange(16, dtype=torch.float32).reshape(4, 4) + 100
    homs = torch.arange(S * 9, dtype=torch.float32).reshape(S, 3, 3) + 1000
    d = torch.arange(S, dtype=torch.float32) + 0.5
    buf = ops.warp_params(homs, k_inv, G, d)
    assert buf.numel() == 32 + 16 * S
    assert torch.equal(buf[0:9], k_inv.reshape(9)) and torch.equal(buf[9:21], G[:3].reshape(12)) and float(buf[21:32].abs().sum()) == 0
    rec = buf[32:].view(S, 16)
    assert torch.equal(rec[:, :9], homs.reshape(S, 9)) and torch.equal(rec[:, 9], d) and float(rec[:, 10:].abs().sum()) == 0
    two = torch.stack([homs, homs + 5000])
    buf2, P = ops.blend_flow_params(k_inv, d, two)
    assert P == 2 and buf2.numel() == 32 + 16 * S * 2
    rec2 = buf2[32:].view(S, 2, 16)
    assert torch.equal(rec2[:, 0, :9], homs.reshape(S, 9)) and torch.equal(rec2[:, 1, :9], (homs + 5000).reshape(S, 9))
    assert torch.equal(rec2[:, 0, 9], d) and torch.equal(rec2[:, 1, 9], d)


def test_sharding_partitions_images():
    from mpiflow_amd import pipeline
    for n, w in [(512, 8), (7, 2), (3, 8), (0, 4)]:
        seen = sorted(i for r in range(w) for i in pipeline.shard_indices(n, r, w))
        assert seen == list(range(n))


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mpiflow_amd import pipeline
dist.init_process_group(backend="gloo")
r, w = dist.get_rank(), dist.get_world_size()
mine = pipeline.shard_indices(11, r, w)
st = pipeline.empty_stats()
for i in mine:
    st = pipeline.merge_stats(st, dict(pairs=1, sum_flow_mag=float(i), hole_px=2.0 * i, kernel_seconds=0.5, max_flow_mag=float(i),
                                      wall_seconds=1.0 + r, neg_min_flow=-float(i)))
tot = pipeline.reduce_stats(st)
if r == 0:
    print("RESULT", tot["pairs"], tot["sum_flow_mag"], tot["hole_px"], tot["max_flow_mag"], tot["wall_seconds"], tot["neg_min_flow"])
dist.destroy_process_group()
'''


def test_stats_all_reduce_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", str(script), ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    pairs, sflow, holes, mx, wall, negmin = map(float, line[1:])
    assert pairs == 11 and sflow == sum(range(11)) and holes == 2 * sum(range(11)) and mx == 10 and wall == 2.0 and negmin == 0.0


_WORKER2 = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from PIL import Image
import numpy as np
from mpiflow_amd import pipeline
dist.init_process_group(backend="gloo")
r, w = dist.get_rank(), dist.get_world_size()
# (1) one process per GPU: distinct indices pass, a shared index is reported by every rank
ids = pipeline.assert_distinct_devices(r)
assert len(set(ids)) == w
try:
    pipeline.assert_distinct_devices(0)
    shared = "not detected"
except RuntimeError as e:
    shared = "detected" if "share a GPU" in str(e) else repr(e)
# (2) mask.max() table: decoded on rank 0 only, broadcast
names = sorted(os.listdir(sys.argv[2]))
if r != 0:
    real_open = Image.open
    def guarded(path, *a, **k):
        raise AssertionError("rank %d decoded %s" % (r, path))
    Image.open = guarded
table = pipeline.mask_max_table(names, sys.argv[2], r, w)
# (3) skip lists and resume counts meet on every rank
merged, resumed = pipeline.gather_reports([("img%d" % r, "reason %d" % r)], r + 1)
dist.barrier()
sys.stdout.write("RESULT %d %s %s %s %s\n" % (r, shared, table, merged, resumed))      # one write per line: the two ranks share the pipe
sys.stdout.flush()
dist.destroy_process_group()
'''


def test_multi_rank_contract_world_size_2_gloo(tmp_path):
    """Two gloo ranks on CPU: the one-process-per-GPU check, the rank-0 mask-max table + broadcast, and the end-of-batch gather of
    skip lists (the N > 1 host logic of gen_3dphoto_dynamic.py; RCCL itself needs the 8-GPU node)."""
    from PIL import Image
    masks = tmp_path / "masks"
    masks.mkdir()
    for n, mx in (("a.png", 3), ("b.png", 0), ("c.png", 7)):
        m = np.zeros((6, 8), np.uint8)
        m[1, 1] = mx
        Image.fromarray(m).save(masks / n)
    (masks / "d.png").write_bytes(b"broken")
    script = tmp_path / "worker2.py"
    script.write_text(_WORKER2)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29642", str(script), ROOT, str(masks)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = sorted(l for l in r.stdout.splitlines() if l.startswith("RESULT"))
    assert len(lines) == 2
    for k, l in enumerate(lines):
        assert l.startswith("RESULT %d detected [3, 0, 7, -1] [('img0', 'reason 0'), ('img1', 'reason 1')] 3" % k), l


def test_flo_round_trip_and_layout(tmp_path):
    from mpiflow_amd import io_formats
    rs = np.random.RandomState(0)
    flow = rs.randn(5, 7, 2).astype(np.float32)
    p = str(tmp_path / "a.flo")
    io_formats.write_flo(p, flow)
    raw = open(p, "rb").read()
    assert len(raw) == 12 + 5 * 7 * 8
    assert np.frombuffer(raw[:4], np.float32)[0] == np.float32(202021.25)
    assert tuple(np.frombuffer(raw[4:12], np.int32)) == (7, 5)            # width then height (write_flow.py:95-96)
    assert np.frombuffer(raw[12:20], np.float32).tolist() == flow[0, 0].tolist()   # u, v interleaved
    assert bits_equal(io_formats.read_flo(p), flow) == 0


def test_async_writer_writes_everything_and_propagates_errors(tmp_path):
    from mpiflow_amd import io_formats
    w = io_formats.AsyncWriter(threads=4, max_pending=3)
    rs = np.random.RandomState(1)
    flows = [rs.randn(6, 9, 2).astype(np.float32) for _ in range(20)]
    for i, f in enumerate(flows):
        w.flo(str(tmp_path / ("f%d.flo" % i)), f)
        w.png_bgr(str(tmp_path / ("i%d.png" % i)), (rs.rand(6, 9, 3) * 255).astype(np.uint8))
    w.close()
    for i, f in enumerate(flows):
        assert bits_equal(io_formats.read_flo(str(tmp_path / ("f%d.flo" % i))), f) == 0
        assert os.path.getsize(tmp_path / ("i%d.png" % i)) > 0
    w2 = io_formats.AsyncWriter(threads=2)
    w2.flo(str(tmp_path / "no_such_dir" / "x.flo"), flows[0])
    with pytest.raises(Exception):
        w2.close()


def test_write_flow_mirror_writes_the_reference_bytes(tmp_path, capsys):
    """mpiflow_amd.write_flow.writeFlow / readFlow == write_flow.py:14-33, :74-103 (file bytes recorded from the reference)"""
    from mpiflow_amd import write_flow
    g = load_golden("flo_file")
    a, b = str(tmp_path / "a.flo"), str(tmp_path / "b.flo")
    write_flow.writeFlow(a, g["flow"])
    write_flow.writeFlow(b, g["flow"][:, :, 0].astype(np.float64), g["flow"][:, :, 1].astype(np.float64))
    assert open(a, "rb").read() == g["file_bytes"].tobytes()
    assert open(b, "rb").read() == g["file_bytes_uv"].tobytes()
    back = write_flow.readFlow(a)
    assert back.dtype == np.float32 and np.array_equal(back, g["read_back"])
    bad = tmp_path / "bad.flo"
    bad.write_bytes(b"\0" * 64)
    assert write_flow.readFlow(str(bad)) is None and "Magic number incorrect" in capsys.readouterr().out


def test_batched_host_algebra_is_bit_identical_to_single_evaluations():
    """The generator builds an image's 2 x repeat poses and a pair's 2 x S homographies in one batched call each; the results
    must be the bits of the one-at-a-time evaluations (which are pinned against the reference)."""
    import random as _random
    from mpiflow_amd import host_math, synth
    g = load_golden("pose_schedule")
    rng = _random.Random(int(g["seed"]))
    params = []
    for _ in range(g["G_dyn"].shape[0]):
        params.append(host_math.draw_pose_parameters(float(g["ext_cz"]), rng=rng))
        params.append(host_math.draw_pose_parameters(float(g["ext_cz"]), base_motions=[0, 0, 0], rng=rng))
    M = host_math.poses_from_parameters(params).numpy()
    assert np.array_equal(M[0::2], g["G_dyn"]) and np.array_equal(M[1::2], g["G_cam"])
    K = torch.from_numpy(synth.intrinsics(48, 64))
    k_inv = host_math.k_inverse(K)
    d = host_math.plane_depths(torch.from_numpy(synth.plane_disparities(20)))
    poses = [torch.from_numpy(g["G_cam"][3]), torch.from_numpy(g["G_dyn"][5]), torch.from_numpy(g["G_dyn"][0])]
    H_ts, H_st = host_math.homographies_multi(poses, k_inv, K, d)
    for i, G in enumerate(poses):
        a, b = host_math.homographies(G, k_inv, K, d)
        assert torch.equal(H_ts[i], a) and torch.equal(H_st[i], b)
    # bench.py's timed host work (--host-prep window): 8 pairs = 16 poses x 64 planes in ONE evaluation, against the per-pose ones
    K = torch.from_numpy(synth.intrinsics(640, 960))
    k_inv = host_math.k_inverse(K)
    d = host_math.plane_depths(torch.from_numpy(synth.plane_disparities(64)))
    rng = _random.Random(7)
    params = []
    for _ in range(8):
        params += [host_math.draw_pose_parameters(0.15, rng=rng), host_math.draw_pose_parameters(0.15, base_motions=(0, 0, 0), rng=rng)]
    rng = _random.Random(7)
    singles = []
    for _ in range(8):
        singles += [host_math.generate_random_pose(0.15, rng=rng), host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)]
    G = host_math.poses_from_parameters(params)
    assert all(torch.equal(G[i], singles[i]) for i in range(16))
    H_ts, H_st = host_math.homographies_multi(list(G), k_inv, K, d)
    for i in range(16):
        a, b = host_math.homographies(singles[i], k_inv, K, d)
        assert torch.equal(H_ts[i], a) and torch.equal(H_st[i], b)


def test_reference_named_entry_points_share_the_generator_cli():
    """gen_3dphoto_dynamic_v2.py is the generator the reference ships (scripts/gen_train_kitti15_v2.sh); gen_coco.sh calls
    gen_3dphoto_dynamic_coco.py.  Both names exist here, take the reference's flags, and the COCO one defaults to the COCO poses."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script in ("gen_3dphoto_dynamic_v2.py", "gen_3dphoto_dynamic_coco.py", "gen_3dphoto_dynamic.py"):
        r = subprocess.run([sys.executable, os.path.join(root, script), "--help"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        for flag in ("--width", "--height", "--seed", "--ext_cz", "--ckpt_path", "--repeat", "--base", "--out"):   # gen_3dphoto_dynamic_v2.py:22-32
            assert flag in r.stdout
    sys.path.insert(0, root)
    import gen_3dphoto_dynamic as g
    assert g.parse(["--base", "b", "--out", "o"]).poses == "v2"
    src = open(os.path.join(root, "gen_3dphoto_dynamic_coco.py")).read()
    assert '"--poses", "coco"' in src


def test_bench_gpus_flag_is_never_silently_ignored():
    """bench.py --gpus N: N > 1 without a launcher spawns the ranks itself and refuses when the box has fewer devices (here: none);
    under a launcher, a WORLD_SIZE that disagrees with --gpus is an error, not a smaller measurement with n_gpus quietly adjusted."""
    bench = os.path.join(ROOT, "bench.py")
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MPIFLOW_FORCE_DEVICE", "MPIFLOW_DIST_BACKEND")}
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=clean)
        assert r.returncode != 0 and "only 0 GPU(s) are visible" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300,
                       env=dict(clean, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "they must agree" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_parameter_blocks_of_all_pairs_in_one_buffer_equal_the_per_block_packing():
    """pipeline.pack_pair_blocks (one strided fill for the 3R blocks of an image) against ops.blend_flow_params / ops.warp_params block by block."""
    import random as _random
    from mpiflow_amd import host_math, ops, pipeline, synth
    for S, R, seed in ((7, 3, 3), (64, 5, 4), (1, 1, 5)):
        K = torch.from_numpy(synth.intrinsics(48, 64))
        d = host_math.plane_depths(torch.from_numpy(synth.plane_disparities(S)))
        k_inv = host_math.k_inverse(K)
        rng = _random.Random(seed)
        poses = [host_math.generate_random_pose(0.15, rng=rng) for _ in range(2 * R)]
        H_ts, H_st = host_math.homographies_multi(poses, k_inv, K, d)
        host, offs, sizes = pipeline.pack_pair_blocks(k_inv, d, H_ts, H_st, poses)
        used = torch.zeros(host.numel(), dtype=torch.bool)
        for r in range(R):
            blocks = [ops.blend_flow_params(k_inv, d, H_ts[2 * r:2 * r + 2])[0]] + [ops.warp_params(H_st[2 * r + v], k_inv, poses[2 * r + v], d) for v in range(2)]
            for j, blk in enumerate(blocks):
                o = offs[r][j]
                assert o % 64 == 0 and blk.numel() == sizes[j]
                assert torch.equal(host[o:o + sizes[j]].view(torch.int32), blk.view(torch.int32)), (S, r, j)
                used[o:o + sizes[j]] = True
        assert not host[~used].any()                      # padding stays zero
