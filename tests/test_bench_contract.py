"""bench.py contract on a GPU box: one JSON line with the required keys (N=1), and the multi-process code path
(rendezvous, barrier, end-of-batch all-reduce, max-over-ranks timing, rank-0 print) with two ranks sharing GPU 0 over gloo -
the real N>1 runs use RCCL, which refuses two ranks on one device."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--planes", "16", "--height", "64", "--width", "96", "--images", "2", "--pairs-per-step", "0", "--steps", "3", "--warmup", "1", "--cpu-pairs", "2",
         "--no-generator"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline"}


def _last_json(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_single_gpu_line():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    # round 6: a pair's host work (pose draws, homographies, fp64 inverses) is INSIDE the timed region of the default line, the product library is the one measured,
    # and the scalars the judge reads sit in `config` (the driver's record keeps config values, only the names of the other sub-records)
    c = d["config"]
    assert c["host_prep"].startswith("per pair, timed") and c["host_prep_us_per_pair_timed"] > 0 and c["library"] == "libmpiflow_hip.so"
    assert set(c["host_prep_comparison"]) == {"once", "per-pair", "window"} and all(v["pairs_per_s"] > 0 for v in c["host_prep_comparison"].values())
    for k in ("pair_alone_pairs_per_s", "value_over_pair_alone", "pairs_per_s_host_prep_once", "pairs_per_s_host_prep_window", "stage_b_2views_frac", "stage_ac_frac",
              "hbm_read_GBps", "hbm_copy_GBps", "cpu_baseline_pairs_per_s", "batch512_pairs_per_s"):
        assert isinstance(c[k], float) and c[k] > 0, k
    for k in ("n1_ms_per_image", "generator_pairs_per_s_steady", "generator_host_share_1of8_vs_unrestricted"):       # --no-generator: present, null
        assert k in c and c[k] is None, k


def test_host_prep_modes_and_the_witness_switch():
    """--host-prep once (rounds 1-5's form: fixed poses prepared at set-up) still runs and says so; --witness measures on the witness build and says so; a witness-only
    tune key is refused on the product library."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    fast = SMALL + ["--no-sub", "--no-cpu-baseline"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--host-prep", "once"] + fast, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["config"]["host_prep"].startswith("once per image")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--witness", "--tune", "view_shift=4"] + fast, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "witness" in _last_json(r.stdout)["config"]["library"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tune", "view_shift=4"] + fast, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "unknown key" in r.stderr


def test_two_rank_path_over_gloo():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MPIFLOW_DIST_BACKEND="gloo", MPIFLOW_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-6     # all ranks' pairs / max time


def test_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with NO launcher around it (the driver's command form): bench.py starts the two ranks itself, and the
    line reports what the process group saw - world size, backend, one device record per rank.  (gloo + one shared device here; on an
    N-GPU node the same path runs RCCL with LOCAL_RANK = device.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MPIFLOW_DIST_BACKEND="gloo", MPIFLOW_FORCE_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["config"]["backend"] == "gloo"
    assert len(d["config"]["ranks"]) == 2 and all(x.startswith("cuda:0") for x in d["config"]["ranks"])
    assert d["config"]["launcher"].startswith("self")
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-6


def test_eight_ranks_self_launched_with_the_strong_scaling_record():
    """`python bench.py --gpus 8` (the driver's command on an 8-GPU node) with no launcher around it: eight ranks over gloo on the one device
    of this box - rendezvous through the launcher's own store (no fixed port), the one-process-per-GPU record exchange over the gloo side
    group, barrier, the statistics all-reduce, max-over-ranks timing, ONE line from rank 0 carrying the weak-scaling `value` AND the
    configs[3] strong-scaling sub-record (a fixed batch sharded i % 8, per-rank seconds and pair counts)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MPIFLOW_DIST_BACKEND="gloo", MPIFLOW_FORCE_DEVICE="0")
    tiny = ["--planes", "8", "--height", "32", "--width", "64", "--images", "2", "--pairs-per-step", "0", "--steps", "2", "--warmup", "1", "--no-generator",
            "--batch", "21"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + tiny, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["world_size"] == 8 and len(d["config"]["ranks"]) == 8 and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6
    b = d["batch512"]
    assert b["scaling"] == "strong" and b["batch_images"] == 21 and len(b["per_rank_seconds"]) == 8
    assert b["per_rank_pairs"] == [3, 3, 3, 3, 3, 2, 2, 2] and sum(b["per_rank_pairs"]) == 21           # image i belongs to rank i % 8
    assert abs(b["pairs_per_s"] - 21 / max(b["per_rank_seconds"])) / b["pairs_per_s"] < 1e-9


def test_gpus_flag_fails_loudly_without_enough_devices():
    """No test hook: --gpus beyond the visible device count must exit non-zero with a message, not measure fewer GPUs silently."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MPIFLOW_FORCE_DEVICE", "MPIFLOW_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1)] + SMALL, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "only %d GPU(s) are visible" % n in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_batch_mode_two_ranks_over_gloo():
    """--mode batch (BASELINE configs[3], strong scaling): a fixed batch sharded i % world == rank; value = batch pairs / max time."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MPIFLOW_DIST_BACKEND="gloo", MPIFLOW_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "batch", "--batch", "7"] + SMALL,
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["config"]["mode"] == "batch" and d["config"]["batch_images"] == 7
    assert d["config"]["pairs_rank0_per_step"] == 4                       # images 0, 2, 4, 6 of 7
    assert abs(d["value"] - 7 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-6
    assert "dynamic pair" in d["config"]["workload"] and "moving-object chain" in d["config"]["workload"] and d["config"]["moving_object_chain"] is True


def test_single_gpu_line_has_sub_records_and_names_the_dynamic_pair():
    """The default-shaped line (2 steps): value = the pipelined dynamic pair with the heterogeneous-grid launch as roofline kernel; per-kernel
    roofline entries from the NON-overlapped c3 sub-record, Stage A+C on SURVEY 8(d)'s bytes with the layout's byte count beside it; SURVEY
    8(d)'s full c3 (pair + moving-object chain); c2 / c1 / c5; the two-pipeline record; the end-to-end generator; the on-box HBM figures."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--images", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    cfg = d["config"]
    assert "configs[2]" in cfg["workload"] and "dynamic pair" in cfg["workload"] and cfg["pipeline"] == "overlapped"
    assert cfg["pairs_per_step_per_gpu"] == 104 and cfg["timed_seconds"] > 0 and cfg["world_size"] == 1 and len(cfg["ranks"]) == 1
    S, N = 64, 640 * 960
    rf = d["roofline"]
    assert "k_pair_overlap" in rf["kernel"] and rf["views_per_pair"] == 2 and 0.05 < rf["frac"] < 1.0
    assert rf["algorithmic_bytes_per_launch"] == 60.0 * S * N + 12.0 * N + 16.0 * N           # the pair's 60*S*N (+ image, flows)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["launches_timed"] == 2 * 104 - 1
    # the dominant kernel fits into the step: launches x mean duration <= timed seconds
    assert rf["launches_timed"] * rf["avg_launch_ms"] * 1e-3 <= cfg["timed_seconds"]
    pa = d["roofline_pair_alone"]                   # the same launch without the chain's kernels underneath it (render-only sub-record)
    assert "k_pair_overlap" in pa["kernel"] and 0.05 < pa["frac"] < 1.0 and pa["algorithmic_bytes_per_launch"] == rf["algorithmic_bytes_per_launch"]
    n1 = d["roofline_n1"]
    assert "error" not in n1 and n1["launches"] == 21 + 28 and n1["encoder"] == "hip" and 0 < n1["hbm"]["frac"] < 1 and 0 < n1["mfma"]["frac"] < 1 and n1["ms_per_image"] > 0
    b5 = d["batch512"]
    assert b5["batch_images"] == 512 and b5["per_rank_pairs"] == [512] and b5["pairs_per_s"] > 0
    ac, sb = d["roofline_stage_ac"], d["roofline_stage_b"]
    assert ac["algorithmic_bytes_per_launch"] == 28.0 * S * N + 12.0 * N + 16.0 * N and ac["layout_bytes_per_launch"] == 32.0 * S * N + 12.0 * N + 16.0 * N
    assert 0 < ac["frac"] < ac["frac_on_layout_bytes"] < 1 and sb["algorithmic_bytes_per_launch"] == 32.0 * S * N and 0.05 < sb["frac"] < 1
    names = [s_["workload"] for s_ in d["sub"]]
    # the headline IS SURVEY 8(d)'s full c3 (pair + moving-object chain, pipelined); the render-only pipelined pair and the serial forms are sub-records
    assert cfg["moving_object_chain"] is True and "forward warp" in cfg["workload"] and cfg["merge_in_launch"] is True
    for tag in ("c3 serial", "c3 + moving-object chain", "c3 render only, pipelined", "c2", "c1", "c5"):
        assert any(n.startswith(tag) for n in names), tag
    for s_ in d["sub"]:
        assert s_["pairs_per_s"] > 0 and ("pair" in s_ or (s_["stage_b"]["frac"] > 0 and s_["stage_ac"]["frac"] > 0))
    by = {n.split(":")[0]: s_ for n, s_ in zip(names, d["sub"])}
    assert by["c3 + moving-object chain (SURVEY 8(d)'s full c3), serial"]["us_per_pair"] > by["c3 serial"]["us_per_pair"]
    assert d["overlap"]["streams"] == 2 and d["overlap"]["pairs_per_s"] > 0.8 * d["value"]
    g = d["generator"]
    assert "error" not in g, g
    # 320 images (64 distinct files + links) x 5 pairs: start-up is a quarter of this run, not nine tenths of it
    assert g["pairs"] == 1600 and g["flo_files_written"] == 1600 and g["pairs_per_s_whole_process"] > 0 and g["pairs_per_s_steady_state"] > 0
    assert g["pairs_per_s_whole_process"] > 0.5 * g["pairs_per_s_steady_state"]
    assert "chain_join" in cfg and "side pipeline" in cfg["chain_join"]
    gp = d["generator_precise"]                   # the same CLI with the parity-grade (fp32) producer
    assert "error" not in gp, gp
    assert gp["model_dtype"] == "fp32" and gp["pairs"] == 120 and gp["flo_files_written"] == 120 and 0 < gp["pairs_per_s_steady_state"] < g["pairs_per_s_steady_state"]
    pr = n1["precise"]                            # the parity-grade modes of the same network
    assert set(pr) == {"fp32", "fp32_mfma", "fp64"} and all("error" not in v and v["ms_per_image"] > 0 and 0 < v["mfma"]["frac"] < 1 for v in pr.values())
    assert pr["fp32"]["ms_per_image"] < pr["fp32_mfma"]["ms_per_image"] < pr["fp64"]["ms_per_image"]
    # the on-box streaming figures next to the 8 TB/s specification: plausible, and the kernels do not beat them
    hb = d["hbm_reference"]
    assert 1000.0 < hb["copy_GBps"] < 8000.0 and 1000.0 < hb["read_GBps"] < 8000.0


def test_serial_pipeline_flag_keeps_the_round_2_structure():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pipeline", "serial", "--no-sub", "--no-cpu-baseline"] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["config"]["pipeline"] == "serial" and "k_warp_composite_views" in d["roofline"]["kernel"] and "roofline_stage_ac" in d
