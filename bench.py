#!/usr/bin/env python3
"""bench.py - image-pairs/sec of the MPI render + flow hot path on MI355X, with roofline and CPU baseline.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload of `value` (BASELINE.json configs[2], the reference's real unit of work - its only entry point always renders
the DYNAMIC pair, utils/utils.py:159-288): 64 planes, 640 x 960, object pose + background pose.  One *step* = every rank
renders `--images` distinct image pairs whose plane stacks are already resident in HBM (synthetic data of the shape
AdaMPI emits; the reference sampler's random poses, fixed per image).  Per pair, three launches:
    mpf_src_blend_flow         Stage A+C: blend the source image into the stack, volume-rendered flow for BOTH poses (P = 2)
                               (+ fused: source frame as uint8 BGR, bilinear tap quads of obj_mask and 1 - obj_mask)
    mpf_warp_composite_views   Stage B, both views in one launch: 64-plane homography warp + front-to-back composite
                               <- dominant / roofline kernel (2 units of 16*S*N algorithmic bytes per launch)
    mpf_merge                  Stage D: thresholds, layer select, uint8 BGR frame, fill mask, merged flow
`--mode batch` (strong scaling, BASELINE configs[3]): a FIXED batch of `--batch` images (default 512) is sharded over the ranks
(i % world == rank, as the generator does) and rendered once per step; `value` = batch pairs / max-over-ranks time.
Images are independent, so ranks share nothing; the only collective is the end-of-batch statistics all-reduce
(RCCL over xGMI under torchrun), issued once after the K timed steps, inside the timed region.

`roofline`: Stage B's algorithmic bytes (16*S*N per view) over its mean launch duration, measured with HIP
events recorded on the launch stream around every Stage B launch inside the timed region; peak 8.0 TB/s.  `roofline.traffic`
is the PMC-measured HBM traffic of that kernel from profiles/roofline_traffic.json - reported only while the kernel source
still has the digest the measurement was taken at (else null: a stale number is worse than none).
`sub`: the other configs on rank 0 at N=1, outside the timed region: camera-only pair (configs[1]), c1 and c5 dynamic
pairs, each with per-kernel roofline entries for Stage B and Stage A+C.
`overlap`: the same dynamic pairs on two HIP streams (own renderers): throughput when Stage A+C of one pair overlaps Stage B of
another - information beside `value`, which stays the single-stream figure so that the roofline entries are clean kernel times.
`hbm_reference`: what a plain device copy / read-only reduction reaches on this box.
`cpu_baseline`: the CPU oracle (our plain-C restatement of the reference algorithm, OpenMP) timed on this host on a
bounded sample of the same workload (dynamic pairs), rank 0, N=1 only.
"""
import argparse
import ctypes
import hashlib
import json
import os
import random
import sys
import time

# The host driver of this node pool only supports dmabuf IPC; with the legacy IPC mode RCCL's peer-memory exchange between the
# per-GPU processes fails (hipIpcGetMemHandle: invalid argument).  The launch environment already exports it; this only
# covers a bare `torchrun bench.py` from a shell that does not.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mpiflow_amd import _lib, host_math, ops, pipeline, synth  # noqa: E402

HBM_PEAK = 8.0e12   # MI355X HBM3E, bytes/s (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", choices=["c3", "c2"], default="c3", help="c3 = dynamic pair (value), c2 = camera-only pair")
    p.add_argument("--mode", choices=["resident", "batch"], default="resident",
                   help="resident: every rank renders --images pairs per step (weak scaling); batch: a fixed --batch images sharded over ranks (strong)")
    p.add_argument("--images", type=int, default=8, help="resident image stacks (pairs per step) per GPU")
    p.add_argument("--batch", type=int, default=512, help="--mode batch: images of the whole job per step (BASELINE configs[3]: 512)")
    p.add_argument("--planes", type=int, default=64)
    p.add_argument("--height", type=int, default=640)
    p.add_argument("--width", type=int, default=960)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-sub", action="store_true", help="skip the c2 / c1 / c5 sub-records")
    p.add_argument("--cpu-pairs", type=int, default=6, help="pairs the CPU oracle renders for cpu_baseline")
    p.add_argument("--sbf-px", type=int, default=0, help="tuning: pixels/thread of Stage A+C (0 = library default)")
    p.add_argument("--single-view-launches", action="store_true", help="tuning: one Stage B launch per view instead of one per pair")
    return p.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here - one process per GPU under
    torch.distributed.run (the form the driver uses itself), LOCAL_RANK = device index, rendezvous on 127.0.0.1 - and pass their
    output and exit status through.  The reference's model is the same: one process per GPU (scripts/gen_train_kitti15_v2.sh:1-4,
    gen_3dphoto_dynamic_v2.py:78).  Fails loudly when the box has fewer devices than ranks asked for."""
    import socket
    import subprocess
    forced = "MPIFLOW_FORCE_DEVICE" in os.environ            # test hook: all ranks on one device over gloo
    have = torch.cuda.device_count()
    if have < a.gpus and not forced:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible on this box" % (a.gpus, have))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MPIFLOW_SELF_LAUNCHED="1")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def init_dist(a):
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree" % (a.gpus, world))
    # test hooks (the multi-process path on a 1-GPU box / in CPU CI): MPIFLOW_DIST_BACKEND=gloo, MPIFLOW_FORCE_DEVICE=0
    backend = os.environ.get("MPIFLOW_DIST_BACKEND", "nccl")
    forced = "MPIFLOW_FORCE_DEVICE" in os.environ
    if forced:
        local = int(os.environ["MPIFLOW_FORCE_DEVICE"])
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants cuda:%d but only %d device(s) are visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    ranks = [pipeline.device_description(local)]
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
        assert dist.get_world_size() == world and dist.get_rank() == rank
        # one process per GPU: checked over a gloo side group BEFORE the first RCCL collective (which would hang on a shared device)
        ids, ranks = pipeline.exchange_device_records(local)
        dup = sorted({i for i in ids if ids.count(i) > 1})
        if dup and backend == "nccl" and not forced:
            raise SystemExit("bench.py: ranks share a GPU: %s" % dict(enumerate(ids)))
    return rank, world, local, backend if world > 1 else None, ranks


def make_image(S, H, W, dev, seed):
    """Synthetic AdaMPI-like stack generated on the device (SURVEY §8(d) distribution)."""
    g = torch.Generator(device=dev).manual_seed(1000 + seed)
    mpi = torch.empty((S, 4, H, W), dtype=torch.float32, device=dev)
    mpi[:, :3] = torch.rand((S, 3, H, W), generator=g, device=dev)
    mpi[:, 3] = torch.relu(3.0 * torch.randn((S, H, W), generator=g, device=dev) - 4.0) + 1e-4
    img = torch.rand((3, H, W), generator=g, device=dev)
    return mpi, img


class Workload:
    """`B` resident images of one shape with fixed random poses; step() renders one pair per image."""

    def __init__(self, S, H, W, B, dev, dynamic, seed0=0, multi_view=True, pose_seed=114514):
        self.S, self.H, self.W, self.B, self.dynamic = S, H, W, B, dynamic
        self.N = H * W
        K, disp = synth.intrinsics(H, W), synth.plane_disparities(S)
        rng = random.Random(pose_seed)
        self.r = pipeline.PairRenderer(S, H, W, dev, n_views=2 if dynamic else 1)
        self.r.multi_view = multi_view
        self.images, self.preps = [], []
        for i in range(B):
            self.images.append(make_image(S, H, W, dev, seed=seed0 + i))
            G_dyn = host_math.generate_random_pose(0.15, rng=rng)                          # utils/utils.py:207
            G_cam = host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)  # :208
            self.preps.append(self.r.prepare(K, disp, [G_cam, G_dyn] if dynamic else [G_dyn]))
        self.om = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev) if dynamic else torch.ones((H, W), dtype=torch.float32, device=dev)
        self.mix = (torch.empty((H, W, 2), dtype=torch.float32, device=dev), torch.empty((H, W, 3), dtype=torch.uint8, device=dev),
                    torch.empty((H, W), dtype=torch.uint8, device=dev))
        self.ev_b, self.ev_ac = [], []

    def pair(self, i, timed):
        r, (mpi, img), prep = self.r, self.images[i], self.preps[i]
        P = prep["P"]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if timed:
            ev[0].record()
        ops.src_blend_flow(mpi, img, out_rgba=r.rgba, out_flows=r.flows[:P], dparams=prep["blend"], P=P, src_u8=r.src_u8,
                           obj_mask=self.om, quads=r.quads[0], quads_complement=r.quads[1] if self.dynamic else None)
        if timed:
            ev[1].record()
            ev[2].record()
        if P > 1 and r.multi_view:
            ops.warp_composite_views(r.rgba, [dict(dparams=prep["warp"][v], quads=r.quads[v], out=r.views[v]) for v in range(P)], interleaved=2)
        else:
            for v in range(P):
                ops.warp_composite(r.rgba, r.quads[v], dparams=prep["warp"][v], out=r.views[v], interleaved=2)
        if timed:
            ev[3].record()
            self.ev_ac.append((ev[0], ev[1]))
            self.ev_b.append((ev[2], ev[3]))
        if self.dynamic:
            v = r.views
            ops.merge(v[0]["rgb"], v[1]["rgb"], v[0]["objmask"], v[1]["objmask"], r.flows[0], r.flows[1], self.om, out=self.mix)

    def step(self, timed, which=None):
        idx = range(self.B) if which is None else which
        for i in idx:
            self.pair(i, timed)
        return len(idx)

    def rooflines(self):
        """Per-kernel roofline entries from the HIP-event brackets collected by timed steps."""
        views = 2 if self.dynamic else 1
        t_b = float(np.mean([a.elapsed_time(b) for a, b in self.ev_b])) * 1e-3
        t_ac = float(np.mean([a.elapsed_time(b) for a, b in self.ev_ac])) * 1e-3
        alg_b = 16.0 * self.S * self.N * views
        # Stage A+C: read 16*S*N + 12*N, write 16*S*N (interleaved stack incl. sigma) + 8*N per pose
        alg_ac = 32.0 * self.S * self.N + 12.0 * self.N + 8.0 * self.N * views
        launches_b = 1 if (views == 1 or self.r.multi_view) else views
        kb = "k_warp_composite_views (Stage B, %d views per launch)" % views if (views > 1 and self.r.multi_view) else "k_warp_composite_v2 (Stage B)"
        return dict(
            stage_b={"bound": "hbm", "kernel": kb, "achieved": alg_b / t_b / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": alg_b / t_b / HBM_PEAK, "algorithmic_bytes_per_launch": alg_b / launches_b,
                     "avg_launch_ms": t_b * 1e3 / launches_b, "launches_timed": len(self.ev_b) * launches_b, "views_per_pair": views},
            stage_ac={"bound": "hbm", "kernel": "k_src_blend_flow (Stage A+C, P=%d)" % views, "achieved": alg_ac / t_ac / 1e9,
                      "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg_ac / t_ac / HBM_PEAK,
                      "algorithmic_bytes_per_launch": alg_ac, "avg_launch_ms": t_ac * 1e3, "launches_timed": len(self.ev_ac)})


def measured_traffic():
    """PMC traffic of the Stage B kernel, valid only for the kernel source it was measured on."""
    tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        rec = json.load(open(tp))
        src = b"".join(open(os.path.join(ROOT, "mpiflow_amd", "csrc", f), "rb").read() for f in ("mpf_render.hip", "mpf_math.h"))
        if rec.get("kernel_source_sha256") == hashlib.sha256(src).hexdigest():
            return rec.get("stage_b_hbm_bytes_per_launch"), rec.get("source")
    except Exception:
        pass
    return None, None


def hbm_reference(dev, nbytes=1 << 30, reps=10):
    """What this box's HBM delivers to the plainest streaming kernels (report an on-box figure beside the 8 TB/s
    spec): mpf_stream_probe's 16-byte-per-lane read-only and copy kernels over 1 GiB, HIP events, outside the timed region."""
    lib = _lib.load()
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timed(mode, moved):
        _lib.check(lib.mpf_stream_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), nbytes, mode, st), "mpf_stream_probe")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.mpf_stream_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), nbytes, mode, st)
        e1.record()
        torch.cuda.synchronize()
        return moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    rec = {"read_GBps": timed(0, nbytes), "copy_GBps": timed(1, 2 * nbytes), "bytes": nbytes,
           "note": "mpf_stream_probe (16 B per lane, non-temporal) on 1 GiB; the roofline fractions above are against the 8 TB/s specification, not against these"}
    del a, b
    return rec


def cpu_baseline(S, H, W, pairs):
    """Time the oracle (checker, used here only as the reported CPU baseline) on `pairs` full dynamic pairs."""
    from oracle import mpi_oracle as orc
    inp = synth.make_inputs(S, H, W, seed=77, kind="white")
    rng = random.Random(114514)
    G_dyn = orc.random_pose(rng, 0.15)
    G_cam = orc.random_pose(rng, 0.15, base_motions=(0, 0, 0))

    def one():
        orc.render_pair(inp["image"], inp["obj_mask"], inp["mpi"], inp["disparity"], inp["K"], G_cam, G_dyn)

    one()
    t0 = time.perf_counter()
    for _ in range(pairs):
        one()
    dt = time.perf_counter() - t0
    return dict(value=pairs / dt, unit="pairs/s", cores=os.cpu_count(), kind="port",
                sample="%d full dynamic pairs (blend + 2 flows, 2 warped views, merge) at %dx%dx%d by the plain-C oracle (OpenMP, %d threads), %.1f s" %
                       (pairs, S, H, W, os.cpu_count(), dt))


def sub_record(name, S, H, W, B, dev, dynamic, steps, multi_view=True):
    w = Workload(S, H, W, B, dev, dynamic, seed0=500, multi_view=multi_view)
    w.step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(steps):
        n += w.step(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = dict(workload=name, pairs_per_s=n / dt, us_per_pair=dt / n * 1e6, pairs_timed=n)
    rec.update(w.rooflines())
    del w
    torch.cuda.empty_cache()
    return rec


def overlap_record(S, H, W, dev, n_streams=2, images=4, steps=8):
    """The same dynamic pairs on `n_streams` HIP streams, each with its own renderer and images: the HBM-bound Stage A+C of one
    pair runs beside the issue-bound Stage B of another.  Reported beside `value`, not as `value`: under overlap a kernel's own
    duration includes the other stream's interference, so the per-kernel roofline entries are taken from the single-stream run."""
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    wls = []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            wls.append(Workload(S, H, W, images, dev, True, seed0=700 + 10 * k))

    def run(n):
        for _ in range(n):
            for i in range(images):
                for wl, st in zip(wls, streams):
                    with torch.cuda.stream(st):
                        wl.pair(i, False)
    run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = steps * images * n_streams
    del wls
    torch.cuda.empty_cache()
    return {"workload": "c3 dynamic pairs on %d streams (separate renderers; Stage A+C of one pair beside Stage B of another)" % n_streams,
            "streams": n_streams, "pairs_per_s": n / dt, "us_per_pair": dt / n * 1e6, "pairs_timed": n}


def main():
    a = parse()
    rank, world, local, backend, rank_devices = init_dist(a)
    dev = torch.device("cuda", local)
    _lib.load()
    if a.sbf_px:
        _lib.check(_lib.load().mpf_tune(b"sbf_px", a.sbf_px))
    S, H, W = a.planes, a.height, a.width
    dynamic = a.workload == "c3"
    if a.mode == "batch":
        mine = pipeline.shard_indices(a.batch, rank, world)          # the generator's sharding: image i belongs to rank i % world
        # each rank keeps min(#mine, --images) distinct stacks resident and cycles through them for its share of the batch
        B = max(1, min(len(mine), a.images))
        order = [j % B for j in range(len(mine))]
    else:
        B = a.images
        order = None
    wl = Workload(S, H, W, B, dev, dynamic, seed0=rank * 1000, multi_view=not a.single_view_launches, pose_seed=114514 + rank)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(a.warmup):
        wl.step(False, order)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = pipeline.empty_stats()
    for _ in range(a.steps):
        st["pairs"] += wl.step(True, order)
    # end-of-batch statistics: the ONE collective of the path (SUM / MAX all-reduce of a 7-float vector, RCCL over xGMI)
    total_pairs = int(pipeline.reduce_stats(st)["pairs"])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    my_pairs = st["pairs"]
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    else:
        total_pairs = my_pairs

    if rank == 0:
        roofs = wl.rooflines()
        traffic, traffic_src = measured_traffic()
        roof = dict(roofs["stage_b"])
        roof["traffic"] = traffic
        if traffic_src:
            roof["traffic_source"] = traffic_src
        cfg_name = ("BASELINE configs[2]: %d planes, %dx%d, full dynamic pair (blend + 2 flows, 2 warped views in one launch, merge)" if dynamic
                    else "BASELINE configs[1]: %d planes, %dx%d, camera-only novel view (blend+flow, warp+composite, u8 frames)") % (S, H, W)
        out = {
            "metric": "image-pairs/sec (+flow) at 640x960x64 planes",
            "value": total_pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if a.mode == "batch" else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg_name, "mode": a.mode,
                       "pairs_per_step_per_gpu": (len(order) if order is not None else B), "resident_stacks_per_gpu": B,
                       "sharding": "independent images per rank (i % world == rank), stats all-reduce only",
                       "device": _lib.device_info(local),
                       "world_size": world, "backend": ("%s (RCCL over xGMI)" % backend if backend == "nccl" else backend) if world > 1 else "none (single rank)",
                       "launcher": "self (bench.py --gpus N spawned torch.distributed.run)" if os.environ.get("MPIFLOW_SELF_LAUNCHED") else
                                   ("external (torchrun)" if world > 1 else "none"),
                       "ranks": rank_devices},
            "roofline": roof,
            "roofline_stage_ac": roofs["stage_ac"],
        }
        if a.mode == "batch":
            out["config"]["batch_images"] = a.batch
            out["config"]["pairs_rank0_per_step"] = len(order)
        del wl
        torch.cuda.empty_cache()
        if world == 1 and not a.no_sub and a.mode == "resident":
            sub = []
            if dynamic:
                sub.append(sub_record("c2: BASELINE configs[1], 64x640x960 camera-only pair", 64, 640, 960, 4, dev, False, 5))
                sub.append(sub_record("c3 with one Stage B launch per view (comparison)", 64, 640, 960, 4, dev, True, 5, multi_view=False))
            else:
                sub.append(sub_record("c3: BASELINE configs[2], 64x640x960 dynamic pair", 64, 640, 960, 4, dev, True, 5))
            sub.append(sub_record("c1: BASELINE configs[0] shape, 32x384x512 dynamic pair (on the GPU: the product has no CPU path)", 32, 384, 512, 8, dev, True, 10))
            sub.append(sub_record("c5: BASELINE configs[4] shape, 128x1024x1536 dynamic pair, random poses", 128, 1024, 1536, 2, dev, True, 5))
            out["sub"] = sub
            if dynamic:
                out["overlap"] = overlap_record(S, H, W, dev)
        if world == 1 and not a.no_sub:
            out["hbm_reference"] = hbm_reference(dev)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, H, W, a.cpu_pairs)
            out["cpu_baseline"]["reference_measured_in_build_container"] = \
                "reference render_3dphoto_dynamic (the same full dynamic pair) 64x640x960: 104.8 s on 8 threads (tests/golden/make_golden.py)"
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
