#!/usr/bin/env python3
"""bench.py - image-pairs/sec of the MPI render + flow hot path on MI355X, with roofline and CPU baseline.

    python bench.py --gpus N --steps K --warmup W          (N > 1: bench.py starts the N ranks itself, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload of `value` (BASELINE.json configs[2], the reference's real unit of work - its only entry point always renders
the DYNAMIC pair, utils/utils.py:159-288): 64 planes, 640 x 960, object pose + background pose.  One *step* = every rank
renders `--pairs-per-step` image pairs, cycling through `--images` distinct plane stacks that are already resident in HBM
(synthetic data of the shape AdaMPI emits; the reference sampler's random poses, fixed per image).  The pairs of a rank form a
two-stage software pipeline (pipeline.OverlappedPairRenderer) - per pair ONE heterogeneous-grid launch and one merge:
    mpf_warp_views_and_blend_next   Stage B of pair i (64-plane homography warp + front-to-back composite of BOTH views) and
                                    Stage A+C of pair i+1 (blend the source image into the stack, volume-rendered flow for both
                                    poses, source frame as uint8 BGR, bilinear tap quads of obj_mask and 1 - obj_mask), their
                                    workgroups interleaved on every CU   <- dominant / roofline kernel: the whole pair's
                                    60*S*N algorithmic bytes per launch
    mpf_merge                       Stage D: thresholds, layer select, uint8 BGR frame, fill mask, merged flow
The very first pair of the timed region pays a stand-alone Stage A+C launch and the very last one a stand-alone Stage B launch
(the pipeline's prologue / epilogue), both inside the timed region: every timed pair is rendered completely.
`--pipeline serial` is the round-2 structure (A+C, Stage B, merge one after the other); it is what the `sub` records use to time
the two kernels on their own.
`--mode batch` (strong scaling, BASELINE configs[3]): a FIXED batch of `--batch` images (default 512) is sharded over the ranks
(i % world == rank, as the generator does) and rendered once per step; `value` = batch pairs / max-over-ranks time.
Images are independent, so ranks share nothing; the only collective is the end-of-batch statistics all-reduce
(RCCL over xGMI), issued once after the K timed steps, inside the timed region.

`roofline`: the dominant kernel's algorithmic bytes per launch (Stage B 16*S*N per view x 2 + Stage A+C
16*S*N + 12*N read and 12*S*N + 8*N per pose written = the pair's 60*S*N) over its mean launch duration, measured with HIP events
recorded on the launch stream around every such launch inside the timed region; peak 8.0 TB/s.  `roofline.traffic` is the
PMC-measured HBM traffic of that kernel from profiles/roofline_traffic.json - reported only while the kernel source still has the
digest the measurement was taken at (else null: a stale number is worse than none).
`roofline_stage_b`, `roofline_stage_ac`: the two stages as kernels of their own (non-overlapped, from the serial c3 sub-record);
Stage A+C on SURVEY 8(d)'s bytes, with the interleaved layout's real byte count beside it.
`sub` (rank 0, N=1, outside the timed region): the serial c3 pair; c3 + the moving-object chain (depth->flow projection, forward warp,
masks on disp = rand: SURVEY 8(d)'s full c3); the camera-only pair (configs[1]); c1 and c5 dynamic pairs.
`overlap`: the same pipelined pairs as two pipelines on two HIP streams (each launch's ramp-down filled by the other stream) -
information beside `value`, which stays the single-stream figure so that the kernel times are clean.
`generator`: the whole data generator (gen_3dphoto_dynamic.py: PNG decode, AdaMPI network on the HIP engine, render, hole filling,
PNG / .flo files) on a small synthetic KITTI-shaped set at 64 x 384 x 1280 - a subprocess, outside the timed region.
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
    p.add_argument("--images", type=int, default=8, help="distinct resident image stacks per GPU")
    p.add_argument("--pairs-per-step", type=int, default=104,
                   help="pairs every rank renders per step, cycling through its resident stacks (104 x 20 steps > 1 s of timed work); 0 = --images")
    p.add_argument("--pipeline", choices=["overlapped", "serial"], default="overlapped",
                   help="overlapped: Stage B of pair i and Stage A+C of pair i+1 in one heterogeneous-grid launch; serial: one kernel after the other")
    p.add_argument("--no-moving-object", action="store_true",
                   help="c3 without the moving-object chain (depth->flow projection, forward warp, masks): the render-only pair of rounds 1-3")
    p.add_argument("--chain-priority", type=int, default=0, help="tuning: 1 = the chain's side stream gets the highest stream priority")
    p.add_argument("--main-priority", type=int, default=-1,
                   help="tuning: run the timed workload on a stream of this priority (-1 = the device's highest) instead of the default stream, so that "
                        "the chain's normal-priority side stream is only dispatched where the pair launches leave room")
    p.add_argument("--main-cu-exclude-stride", type=int, default=0,
                   help="tuning: the pair stream is barred from every n-th compute unit (use with --chain-cu-stride n: the chain then owns those CUs)")
    p.add_argument("--chain-cu-stride", type=int, default=0, help="tuning: the chain's side stream may only use every n-th compute unit (0 = all)")
    p.add_argument("--chain-sides", type=int, default=2, help="side streams the moving-object chains alternate over (2: a chain may take two pair launches before it delays anything)")
    p.add_argument("--merge-in-launch", type=int, default=1,
                   help="1 = Stage D of pair i is a per-pixel prologue of the Stage A+C role of launch i+2 (one launch per pair); 0 = a launch of its own after every pair launch")
    p.add_argument("--chain-ordered", type=int, default=0,
                   help="1 = the chain's results are stream-ordered on the main stream at every pair (event record + wait per pair); 0 = independent side pipeline, joined at the end")
    p.add_argument("--host-prep", choices=["window", "per-pair", "once"], default="window",
                   help="host work of a pair (utils/utils.py:207-208 pose draws + homography_sampler.py:105-122 per-plane homographies and their fp64 inverse): "
                        "window = INSIDE the timed region, every timed pair draws its own two poses, the math batched over the in-flight window of --images pairs "
                        "(one batched evaluation + one pinned upload per window); per-pair = inside the timed region, one prepare() call per pair; "
                        "once = outside the timed region, fixed poses per image (rounds 1-5)")
    p.add_argument("--no-generator", action="store_true", help="skip the end-to-end generator record")
    p.add_argument("--batch", type=int, default=512, help="--mode batch: images of the whole job per step (BASELINE configs[3]: 512)")
    p.add_argument("--planes", type=int, default=64)
    p.add_argument("--height", type=int, default=640)
    p.add_argument("--width", type=int, default=960)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-sub", action="store_true", help="skip the c2 / c1 / c5 sub-records")
    p.add_argument("--cpu-pairs", type=int, default=64, help="most pairs the CPU oracle renders for cpu_baseline (it stops after ~12 s of CPU work)")
    p.add_argument("--sbf-px", type=int, default=0, help="tuning: pixels/thread of Stage A+C (0 = library default)")
    p.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                   help="tuning: mpf_tune(KEY, INT) before anything runs (chain_prio, chain_grid, conv_pf, ...); recorded in config.tune")
    p.add_argument("--witness", action="store_true",
                   help="run on libmpiflow_hip_witness.so (the -DMPF_WITNESS build: retired kernel variants and timing ablations, --tune ovl_xcd_a / view_shift / "
                        "ovl_ablate / stage_b ...); recorded in config.library - a line measured on it is not the product's")
    p.add_argument("--single-view-launches", action="store_true", help="tuning: one Stage B launch per view instead of one per pair")
    return p.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here - one process per GPU under
    torch.distributed.run (the form the driver uses itself), LOCAL_RANK = device index, rendezvous on 127.0.0.1 - and pass their
    output and exit status through.  The reference's model is the same: one process per GPU (scripts/gen_train_kitti15_v2.sh:1-4,
    gen_3dphoto_dynamic_v2.py:78).  Fails loudly when the box has fewer devices than ranks asked for."""
    import subprocess
    forced = "MPIFLOW_FORCE_DEVICE" in os.environ            # test hook: all ranks on one device over gloo
    have = torch.cuda.device_count()
    if have < a.gpus and not forced:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible on this box" % (a.gpus, have))
    # --standalone: the launcher's own c10d store picks a free port and keeps it (no bind-then-close race with other jobs of the box);
    # --local-addr: the container's host name may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           os.path.abspath(__file__)] + sys.argv[1:]
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


def alg_bytes(S, N, views):
    """SURVEY.md 8(d) algorithmic bytes (fp32 planar API layout): Stage B 16*S*N per view; Stage A+C reads 16*S*N + 12*N and writes
    12*S*N + 8*N per pose.  The interleaved layout really moves 16*S*N on the write side (sigma is re-written): reported beside it."""
    b = 16.0 * S * N * views
    ac = 16.0 * S * N + 12.0 * N + 12.0 * S * N + 8.0 * N * views
    ac_layout = 32.0 * S * N + 12.0 * N + 8.0 * N * views
    return b, ac, ac_layout


class Workload:
    """`B` resident images of one shape with fixed random poses, rendered ONE KERNEL AFTER THE OTHER (the round-2 structure): per pair
    Stage A+C, Stage B (both views in one launch), merge.  Used for the per-kernel roofline entries and the comparison records."""

    def __init__(self, S, H, W, B, dev, dynamic, seed0=0, multi_view=True, pose_seed=114514, moving_object=False):
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
        self.mo, self.mo_disp = make_moving_object_chain(H, W, K, dev, seed0) if moving_object else (None, None)

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
        if self.mo is not None:
            self.mo.run(self.mo_disp, self.om, r.src_u8)

    def step(self, timed, which=None):
        idx = range(self.B) if which is None else which
        for i in idx:
            self.pair(i % self.B, timed)
        return len(idx)

    def finish(self):
        return None

    def rooflines(self):
        """Per-kernel roofline entries from the HIP-event brackets collected by timed steps."""
        views = 2 if self.dynamic else 1
        t_b = float(np.mean([a.elapsed_time(b) for a, b in self.ev_b])) * 1e-3
        t_ac = float(np.mean([a.elapsed_time(b) for a, b in self.ev_ac])) * 1e-3
        alg_b, alg_ac, lay_ac = alg_bytes(self.S, self.N, views)
        launches_b = 1 if (views == 1 or self.r.multi_view) else views
        kb = "k_warp_composite_views (Stage B, %d views per launch)" % views if (views > 1 and self.r.multi_view) else "k_warp_composite_v2 (Stage B)"
        return dict(
            stage_b={"bound": "hbm", "kernel": kb, "achieved": alg_b / t_b / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": alg_b / t_b / HBM_PEAK, "algorithmic_bytes_per_launch": alg_b / launches_b,
                     "avg_launch_ms": t_b * 1e3 / launches_b, "launches_timed": len(self.ev_b) * launches_b, "views_per_pair": views},
            stage_ac={"bound": "hbm", "kernel": "k_src_blend_flow (Stage A+C, P=%d)" % views, "achieved": alg_ac / t_ac / 1e9,
                      "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg_ac / t_ac / HBM_PEAK,
                      "algorithmic_bytes_per_launch": alg_ac, "avg_launch_ms": t_ac * 1e3, "launches_timed": len(self.ev_ac),
                      "layout_bytes_per_launch": lay_ac, "frac_on_layout_bytes": lay_ac / t_ac / HBM_PEAK,
                      "note": "algorithmic bytes per SURVEY 8(d) (read 16SN + 12N, write 12SN + 8N per pose); the interleaved RGBA stack Stage B "
                              "gathers from re-writes sigma, so the kernel really moves layout_bytes_per_launch"})


def make_moving_object_chain(H, W, K, dev, seed):
    """SURVEY 8(d)'s c3 adds "forward-warp on disp = rs.rand(H, W)": the moving-object chain of moving_obj.py:29-150 on device-resident
    inputs (mpiflow_amd.moving_obj.MovingObjectChain: projection fused into the forward splat's first sort pass, splat, masks - one C call,
    3 launches), a fixed object pose of the reference's magnitude (moving_obj.py:81-98; the angles are zeroed there).  -> (chain, disp)"""
    from mpiflow_amd import moving_obj
    g = torch.Generator(device=dev).manual_seed(4242 + seed)
    disp = torch.rand((H, W), generator=g, device=dev)
    K3 = torch.from_numpy(np.asarray(K, dtype=np.float32)).reshape(3, 3)
    Ti = host_math.transformation_from_parameters(torch.zeros(1, 1, 3), torch.tensor([[0.07, -0.06, 0.08]]))
    return moving_obj.MovingObjectChain(H, W, K3, torch.inverse(K3.double()).float(), dev, T_obj=Ti, n_buffers=6), disp


class PipelinedWorkload:
    """The same pairs as Workload(dynamic=True), rendered by pipeline.OverlappedPairRenderer: Stage B of pair i and Stage A+C of pair
    i+1 in one heterogeneous-grid launch.  finish() flushes the pipeline (the last pair's stand-alone Stage B)."""
    dynamic = True

    def __init__(self, S, H, W, B, dev, seed0=0, pose_seed=114514, moving_object=False, chain_priority=False, chain_ordered=False, merge_in_launch=True, chain_cu_stride=0, chain_sides=2,
                 host_prep="once"):
        self.S, self.H, self.W, self.B, self.N = S, H, W, B, H * W
        K, disp = synth.intrinsics(H, W), synth.plane_disparities(S)
        rng = random.Random(pose_seed)
        # host_prep != "once": the pose draws and the per-plane homographies of EVERY pair are host work inside step() - what the reference does per pair
        # (utils/utils.py:207-208, homography_sampler.py:105-122); the set-up poses below then only serve the warm-up of the first window
        self.host_prep, self.K, self.disp, self.rng = host_prep, torch.as_tensor(np.asarray(K, np.float32)), torch.as_tensor(np.asarray(disp, np.float32)), rng
        self.host_seconds = 0.0
        self.copy_stream = torch.cuda.Stream(dev)
        # merge_in_launch: Stage D of pair i rides in launch i+2 (per-pixel prologue of its Stage A+C role): ONE launch per pair, nothing between
        self.r = pipeline.OverlappedPairRenderer(S, H, W, dev, merge_in_launch=merge_in_launch)
        # SURVEY 8(d)'s full c3: the moving-object chain of every pair runs on the renderer's SIDE stream, issued right behind the launch whose
        # Stage A+C role wrote the pair's uint8 source frame and handed back with the pair one launch later (OverlappedPairRenderer.attach_chain)
        self.mo, self.mo_disp = make_moving_object_chain(H, W, K, dev, seed0) if moving_object else (None, None)
        if moving_object:
            # the chain as an independent side pipeline (attach_chain(ordered=False)): nothing of it is inserted into the main stream; the
            # inputs are resident since set-up (ready event recorded once), finish() joins the side stream inside the timed region
            self.r.attach_chain(self.mo, high_priority=chain_priority, ordered=chain_ordered, cu_stride=chain_cu_stride, sides=chain_sides)
        self.images, self.preps = [], []
        for i in range(B):
            self.images.append(make_image(S, H, W, dev, seed=seed0 + i))
            G_dyn = host_math.generate_random_pose(0.15, rng=rng)                          # utils/utils.py:207
            G_cam = host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)  # :208
            self.preps.append(self.r.prepare(K, disp, [G_cam, G_dyn]))
        self.om = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev)
        # a pair's outputs are written up to two push() calls after it was enqueued: three sets alternate
        self.mix = [(torch.empty((H, W, 2), dtype=torch.float32, device=dev), torch.empty((H, W, 3), dtype=torch.uint8, device=dev),
                     torch.empty((H, W), dtype=torch.uint8, device=dev)) for _ in range(3)]
        self.n_pushed = 0
        self.ev = []
        self.timed = False
        self.inputs_ready = torch.cuda.Event()
        self.inputs_ready.record()                                   # everything the chain reads (images, disparity, mask) is resident from here on

        def hook(launch):
            if not self.timed:
                return launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            self.ev.append((e0, e1))
        self.r.on_fused = hook

    def _draw_window(self, n):
        """The host work of `n` pairs the way the reference orders it - per pair the dynamic pose, then the camera pose (utils/utils.py:207-208, Python's
        `random` stream) - with the arithmetic batched over the window: ONE transformation_from_parameters over the 2n poses, ONE homography / fp64-inverse
        evaluation over their 2n x S matrices, ONE pinned buffer and H2D copy for the 3n parameter blocks (pipeline.prepare_many; bit-identical to n
        prepare() calls - tests/test_host_logic.py).  -> n prep dicts"""
        t0 = time.perf_counter()
        params = []
        for _ in range(n):
            dyn = host_math.draw_pose_parameters(0.15, rng=self.rng)
            cam = host_math.draw_pose_parameters(0.15, base_motions=(0, 0, 0), rng=self.rng)
            params += [cam, dyn]                                     # prepare()'s order: view 0 = camera pose (samples obj_mask), view 1 = dynamic pose
        G = host_math.poses_from_parameters(params)
        # the window's ONE upload goes on a copy stream: the host runs several launches ahead of the GPU, so the copy is done long before the pair stream
        # reaches the event it waits for - in the pair stream itself it would sit between two pair launches
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.copy_stream):
            preps = self.r.prepare_many(self.K, self.disp, [[G[2 * r], G[2 * r + 1]] for r in range(n)])
            ev = torch.cuda.Event()
            ev.record()
        preps[0]["blend"].record_stream(main)                          # (ONE device buffer behind all 3n blocks: allocated on the copy stream, read on the pair stream)
        main.wait_event(ev)
        self.host_seconds += time.perf_counter() - t0
        return preps

    def step(self, timed, which=None):
        idx = list(range(self.B) if which is None else which)
        self.timed = timed
        moving = (self.mo_disp, self.om) if self.mo is not None else None
        win = {"once": 0, "per-pair": 1, "window": self.B}[self.host_prep]
        preps = None
        for k, i in enumerate(idx):
            mpi, img = self.images[i % self.B]
            if win == 0:
                prep = self.preps[i % self.B]
            elif win == 1:
                t0 = time.perf_counter()
                G_dyn = host_math.generate_random_pose(0.15, rng=self.rng)                          # utils/utils.py:207
                G_cam = host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=self.rng)  # :208
                prep = self.r.prepare(self.K, self.disp, [G_cam, G_dyn])
                self.host_seconds += time.perf_counter() - t0
            else:
                if k % win == 0:
                    preps = self._draw_window(min(win, len(idx) - k))
                prep = preps[k % win]
            self.r.push(mpi, img, prep, self.om, out=self.mix[self.n_pushed % 3], moving=moving, moving_ready=self.inputs_ready if moving else None)
            self.n_pushed += 1
        return len(idx)

    def finish(self):
        self.r.flush()

    def rooflines(self):
        t = float(np.mean([a.elapsed_time(b) for a, b in self.ev])) * 1e-3
        alg_b, alg_ac, lay_ac = alg_bytes(self.S, self.N, 2)
        return dict(pair={"bound": "hbm", "kernel": "k_pair_overlap (Stage B of pair i, 2 views + Stage A+C of pair i+1, one heterogeneous grid)",
                          "achieved": (alg_b + alg_ac) / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": (alg_b + alg_ac) / t / HBM_PEAK,
                          "algorithmic_bytes_per_launch": alg_b + alg_ac, "avg_launch_ms": t * 1e3, "launches_timed": len(self.ev), "views_per_pair": 2,
                          "layout_bytes_per_launch": alg_b + lay_ac,
                          "note": "algorithmic bytes = the whole dynamic pair per SURVEY 8(d): 2 x 16SN (Stage B) + 16SN + 12N read + 12SN + 16N written (Stage A+C)"})


def measured_traffic(kind="pair"):
    """PMC traffic of the dominant kernel ("pair": k_pair_overlap, "stage_b": k_warp_composite_views), valid only for the kernel
    source it was measured on."""
    tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        rec = json.load(open(tp))
        src = b"".join(open(os.path.join(ROOT, "mpiflow_amd", "csrc", f), "rb").read() for f in ("mpf_render.hip", "mpf_math.h"))
        if rec.get("kernel_source_sha256") == hashlib.sha256(src).hexdigest():
            key = "pair_hbm_bytes_per_launch" if kind == "pair" else "stage_b_hbm_bytes_per_launch"
            return rec.get(key), rec.get("source")
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


def cpu_baseline(S, H, W, pairs, budget_s=12.0, chain=True):
    """Time the oracle (checker, used here only as the reported CPU baseline) on the workload of `value`: full dynamic pairs + the
    moving-object chain of every pair (SURVEY 8(d)'s c3) - a bounded sample: pairs are rendered until `budget_s` seconds of CPU work have
    been spent (at least 2, at most `pairs`)."""
    from oracle import mpi_oracle as orc
    inp = synth.make_inputs(S, H, W, seed=77, kind="white")
    rng = random.Random(114514)
    G_dyn = orc.random_pose(rng, 0.15)
    G_cam = orc.random_pose(rng, 0.15, base_motions=(0, 0, 0))
    disp = np.random.RandomState(4242).rand(H, W).astype(np.float32)
    inv_K = np.linalg.inv(np.asarray(inp["K"], np.float64)).astype(np.float32)
    T_obj = host_math.transformation_from_parameters(torch.zeros(1, 1, 3), torch.tensor([[0.07, -0.06, 0.08]]))[0].numpy()

    def one():
        r = orc.render_pair(inp["image"], inp["obj_mask"], inp["mpi"], inp["disparity"], inp["K"], G_cam, G_dyn)
        if chain:
            orc.moving_object(disp, r["src_np"], inp["K"], inv_K, inp["obj_mask"], T_obj)

    one()
    t0 = time.perf_counter()
    n = 0
    while n < max(2, pairs) and (n < 2 or time.perf_counter() - t0 < budget_s):
        one()
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="pairs/s", cores=os.cpu_count(), kind="port",
                sample="%d full dynamic pairs (blend + 2 flows, 2 warped views, merge%s) at %dx%dx%d by the plain-C oracle (OpenMP, %d threads), %.1f s" %
                       (n, " + the moving-object chain: depth->flow projection, serial forward warp, masks" if chain else "", S, H, W, os.cpu_count(), dt))


def sub_record(name, S, H, W, B, dev, dynamic, steps, multi_view=True, pipelined=False, moving_object=False, host_prep="once"):
    """One workload outside the timed region (rank 0, N=1): pairs/s plus per-kernel roofline entries from HIP-event brackets."""
    if pipelined:
        w = PipelinedWorkload(S, H, W, B, dev, seed0=500, moving_object=moving_object, host_prep=host_prep)
    else:
        w = Workload(S, H, W, B, dev, dynamic, seed0=500, multi_view=multi_view, moving_object=moving_object)
    w.step(False)
    w.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for _ in range(steps):
        n += w.step(True)
    w.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = dict(workload=name, pairs_per_s=n / dt, us_per_pair=dt / n * 1e6, pairs_timed=n)
    rec.update(w.rooflines())
    del w
    torch.cuda.empty_cache()
    return rec


def overlap_record(S, H, W, dev, n_streams=2, images=4, steps=8):
    """The pipelined dynamic pairs as `n_streams` pipelines on as many HIP streams, each with its own renderer and images: while one
    stream's heterogeneous-grid launch drains, the other's fills the freed slots.  Reported beside `value`, not as `value`: under
    stream-level overlap a kernel's own duration includes the other stream's interference, so the roofline entries come from the
    single-stream runs."""
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    wls = []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            wls.append(PipelinedWorkload(S, H, W, images, dev, seed0=700 + 10 * k))

    def run(n):
        for _ in range(n):
            for i in range(images):
                for wl, st in zip(wls, streams):
                    with torch.cuda.stream(st):
                        wl.step(False, [i])
        for wl, st in zip(wls, streams):
            with torch.cuda.stream(st):
                wl.finish()
    run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = steps * images * n_streams
    del wls
    torch.cuda.empty_cache()
    return {"workload": "c3 dynamic pairs, pipelined (Stage B of pair i + Stage A+C of pair i+1 per launch), %d pipelines on %d HIP streams" % (n_streams, n_streams),
            "streams": n_streams, "pairs_per_s": n / dt, "us_per_pair": dt / n * 1e6, "pairs_timed": n}


def generator_record(n_images=320, repeat=5, timeout=900, n_distinct=64, model_dtype="auto", cpus=None, writers=None):
    """The data generator end to end (gen_3dphoto_dynamic.py, the reference's entry point gen_3dphoto_dynamic_v2.py:20-122): PNG decode,
    input stage, AdaMPI network (random weights of the reference's architecture: no checkpoint offline) on the HIP engine, blend once per
    image, `repeat` pairs per image, hole filling (cv2.inpaint's NS restated, on the writer threads), PNG + .flo files - on a synthetic
    KITTI-shaped set (375 x 1242 PNGs -> 64 planes x 384 x 1280, the reference's defaults).  A subprocess, outside the timed region."""
    import shutil
    import subprocess
    import tempfile
    from PIL import Image
    tmp = tempfile.mkdtemp(prefix="mpf_gen_")
    try:
        base = os.path.join(tmp, "data")
        for d in ("images", "disps", "masks"):
            os.makedirs(os.path.join(base, d))
        rs = np.random.RandomState(0)
        yy, xx = np.mgrid[0:375, 0:1242]
        for i in range(min(n_images, n_distinct)):
            img = (np.clip(0.5 + 0.25 * np.sin(xx / (17.0 + i)) + 0.25 * np.cos(yy / 23.0) + 0.05 * rs.randn(375, 1242), 0, 1) * 255).astype(np.uint8)
            Image.fromarray(np.stack([img, np.roll(img, 7, 1), np.roll(img, 13, 0)], -1)).save(os.path.join(base, "images", "%04d.png" % i))
            Image.fromarray((255 * (0.1 + 0.8 * yy / 375)).astype(np.uint8)).save(os.path.join(base, "disps", "%04d.png" % i))
            m = np.zeros((375, 1242), np.uint8)
            m[150:300, 300:600] = 1
            m[200:330, 800:1000] = 2
            Image.fromarray(m).save(os.path.join(base, "masks", "%04d.png" % i))
        for i in range(n_distinct, n_images):                # the rest of the set: links to the distinct files (decoded, uploaded and rendered like any other image)
            for d in ("images", "disps", "masks"):
                os.symlink(os.path.join(base, d, "%04d.png" % (i % n_distinct)), os.path.join(base, d, "%04d.png" % i))
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        cmd = [sys.executable, os.path.join(ROOT, "gen_3dphoto_dynamic.py"), "--base", base, "--out", os.path.join(tmp, "out"), "--repeat", str(repeat),
               "--mpi-from", "model", "--ckpt_path", "random:0", "--model-engine", "hip", "--model-dtype", model_dtype, "--inpaint", "builtin"]
        if writers:
            cmd += ["--writers", str(writers)]
        # cpus: the process (decode / writer / submitting threads alike) is confined to that many logical CPUs - one rank's share of a node's host
        pre = (lambda: os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:cpus]))) if cpus else None
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, preexec_fn=pre)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}
        steady = [l for l in r.stdout.splitlines() if l.startswith("steady state")]
        startup = [l for l in r.stdout.splitlines() if l.startswith("start-up:")]
        summary = [l for l in r.stdout.splitlines() if l.startswith("pairs ")]
        wline = [l for l in r.stdout.splitlines() if l.startswith("writers:")]
        n_files = len(os.listdir(os.path.join(tmp, "out", "flows")))
        rec = {"workload": "gen_3dphoto_dynamic.py end to end: %d synthetic 375x1242 images -> 64 planes x 384 x 1280, repeat %d, AdaMPI (random weights) on the HIP "
                           "engine, NS hole filling on the writer threads, PNG + .flo written" % (n_images, repeat),
               "producer_precision": "HIP engine: fp16 storage, fp16 MFMA, fp32 accumulate and epilogue - the precision of the reference's own GPU run (.half(), "
                                     "gen_3dphoto_dynamic_v2.py:46,59,82-84), NOT the fp32 CPU parity target of the render path; its error against the fp32 model "
                                     "(random weights) is bounded by tests/test_conv_engine.py ENGINE_BARS, e.g. mean |sigmoid(rgb)| 3.2e-3 at this size "
                                     "(torch fp16 autocast: 1.2e-2); --model-engine hip --model-dtype fp32 runs the parity-grade engine (roofline_n1.precise)",
               "model_dtype": model_dtype, "n_images": n_images, "n_distinct": min(n_images, n_distinct), "repeat": repeat, "pairs": n_images * repeat, "flo_files_written": n_files, "process_seconds": dt,
               "pairs_per_s_whole_process": n_images * repeat / dt, "summary_line": summary[-1] if summary else None}
        if wline:
            rec["writer_stages"] = wline[-1]
        if startup:
            rec["startup_line"] = startup[-1]
            rec["startup_seconds"] = float(startup[-1].split(":")[1].split("s")[0])
            rec["whole_process_note"] = ("a %d-pair run is ~%.1f s of steady state behind a fixed start-up; the whole-process rate approaches the steady "
                                         "state as 1 / (1 + start-up / run time)" % (n_images * repeat, n_images * repeat / max(1e-9, float(steady[-1].split(":")[1].split("pairs/s")[0])) if steady else float("nan")))
        if steady:
            rec["pairs_per_s_steady_state"] = float(steady[-1].split(":")[1].split("pairs/s")[0])
            rec["steady_state_note"] = "the CLI's own figure: pairs after the first image / time after the first image (start-up = graph capture and first launches, excluded)"
        return rec
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


MFMA_F16_PEAK = 2.5e15   # dense fp16 MFMA, flop/s (MI355X_MICROARCH.md; AMD's headline figure includes 2:1 sparsity and is not used)


def n1_record(dev, S=64, H=384, W=1280, iters=10):
    """SURVEY 8(f) N1 - the AdaMPI producer on the HIP engine (mpiflow_amd.model.engine.HipPredictor, random weights of the reference's
    architecture, the generator's 64 x 384 x 1280): one image = 20 mpf_conv3x3_f16 launches + mpf_plane_masks + the single-image encoder / bottleneck
    (24 mpf_conv2d_f32 launches, 3 max-pools, the input normalisation: fp32, on a side stream), replayed from one hipGraph.  Algorithmic flops = the reference's own convolutions on the real channel counts; algorithmic
    bytes = every layer's sources read once and its output written once in the engine's storage types (HipPredictor.accounting).  Both
    roofline fractions are of the whole forward: it is bound by neither alone."""
    from mpiflow_amd.model import MPIPredictor
    from mpiflow_amd.model.engine import HipPredictor
    m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
    hp = HipPredictor(m, encoder_dtype=None, graph=True)
    g = torch.Generator(device=dev).manual_seed(5)
    img, dsp = torch.rand((1, 3, H, W), generator=g, device=dev), torch.rand((1, 1, H, W), generator=g, device=dev)
    for _ in range(2):
        hp(img, dsp)                                                            # capture (every layer records the shapes of its launch) + one replay
    torch.cuda.synchronize()
    rows, tot = hp.accounting()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hp(img, dsp)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    rec = {"workload": "AdaMPI producer (N1) on the HIP engine: %d planes x %d x %d, fp16 storage / fp32 accumulate, one hipGraph replay per image" % (S, H, W),
           "ms_per_image": t * 1e3, "images_per_s": 1.0 / t, "launches": sum(r.get("launches", 1) for r in rows), "encoder": hp.encoder_kind,
           "algorithmic_flops_per_image": tot["flops"], "algorithmic_bytes_per_image": tot["bytes"],
           "hbm": {"bound": "hbm", "achieved": tot["bytes"] / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": tot["bytes"] / t / HBM_PEAK},
           "mfma": {"bound": "mfma", "achieved": tot["flops"] / t / 1e12, "peak": MFMA_F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": tot["flops"] / t / MFMA_F16_PEAK},
           "layers": [{"name": r["name"], "GB": r["bytes"] / 1e9, "GFLOP": r["flops"] / 1e9} for r in rows]}
    del hp
    torch.cuda.empty_cache()
    # the PARITY-GRADE modes of the same network (mpiflow_amd.model.precise.PrecisePredictor: every convolution on mpf_pconv; materialised fp32 / fp64 NHWC
    # activations; eager launches).  fp32 = what --model-dtype fp32 runs: products from bf16 pieces on the matrix cores (six bf16 MFMA flops per algorithmic
    # flop, so its ceiling is a sixth of the dense bf16 rate); fp32_mfma / fp64: the dense fp32 / fp64 MFMA rates (157 / 79 TFLOP/s).
    from mpiflow_amd.model.precise import PrecisePredictor
    rec["precise"] = {}
    for name, dt, x3, peak in (("fp32", torch.float32, True, MFMA_F16_PEAK / 6), ("fp32_mfma", torch.float32, False, 157.3e12), ("fp64", torch.float64, False, 78.6e12)):
        try:
            pp = PrecisePredictor(m, dtype=dt, x3=x3)
            pp(img, dsp)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(2):
                pp(img, dsp)
            e1.record()
            torch.cuda.synchronize()
            tp = e0.elapsed_time(e1) / 2 * 1e-3
            _, ptot = pp.accounting()
            what = {"fp32": "fp32 tensors, products from the three bf16 pieces of each factor on v_mfma_f32_16x16x32_bf16, fp32 blocks of 64 products carried in fp64",
                    "fp32_mfma": "fp32 tensors, products on v_mfma_f32_16x16x4_f32, fp32 blocks of 64 products carried in fp64", "fp64": "fp64 throughout"}[name]
            rec["precise"][name] = {"workload": "the same image on the parity-grade engine: %s (tests/test_precise_engine.py: fp64 = the torch modules in double to 1e-10; both fp32 "
                                                "forms closer to them than torch's own fp32)" % what, "ms_per_image": tp * 1e3, "launches": len(pp.layers()) + 20,
                                    "algorithmic_flops_per_image": ptot["flops"], "materialised_bytes_per_image": ptot["bytes"],
                                    "mfma": {"bound": "mfma", "achieved": ptot["flops"] / tp / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ptot["flops"] / tp / peak}}
            del pp
        except Exception as e:                                                   # noqa: BLE001
            rec["precise"][name] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    del m
    torch.cuda.empty_cache()
    return rec


def main():
    a = parse()
    rank, world, local, backend, rank_devices = init_dist(a)
    dev = torch.device("cuda", local)
    _lib.load()
    # the product's host math is a few hundred 3x3 / 4x4 matrices per pair: torch's intra-op thread pool only costs there (a batched 1024 x 3 x 3 inverse fanned out over
    # a 256-thread host took 10x its single-thread time) - one thread, as gen_3dphoto_dynamic.py runs it; restored for the CPU baseline below
    host_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    if a.witness:
        _lib.select_witness()
    if a.sbf_px:
        _lib.check(_lib.load().mpf_tune(b"sbf_px", a.sbf_px))
    for kv in a.tune:
        key, val = kv.split("=")
        _lib.check(_lib.load().mpf_tune(key.encode(), int(val)), "mpf_tune(%s)" % kv)
    S, H, W = a.planes, a.height, a.width
    dynamic = a.workload == "c3"
    pipelined = dynamic and a.pipeline == "overlapped" and not a.single_view_launches
    if a.mode == "batch":
        mine = pipeline.shard_indices(a.batch, rank, world)          # the generator's sharding: image i belongs to rank i % world
        # each rank keeps min(#mine, --images) distinct stacks resident and cycles through them for its share of the batch
        B = max(1, min(len(mine), a.images))
        order = list(range(len(mine)))
    else:
        B = a.images
        order = list(range(a.pairs_per_step if a.pairs_per_step > 0 else B))
    chain = dynamic and not a.no_moving_object
    main_stream = torch.cuda.Stream(dev, priority=a.main_priority) if a.main_priority else None
    masked_main = None
    if a.main_cu_exclude_stride > 1:                         # tuning: the pair stream may use every CU EXCEPT those the chain's side stream is confined to
        masked_main = ctypes.c_void_p()
        _lib.check(_lib.load().mpf_stream_create_cu_subset(-a.main_cu_exclude_stride, 0, ctypes.byref(masked_main)), "mpf_stream_create_cu_subset")
        main_stream = torch.cuda.ExternalStream(masked_main.value, device=dev)
    if main_stream is not None:
        main_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main_stream)
    if pipelined:
        wl = PipelinedWorkload(S, H, W, B, dev, seed0=rank * 1000, pose_seed=114514 + rank, moving_object=chain, chain_priority=bool(a.chain_priority), chain_ordered=bool(a.chain_ordered), merge_in_launch=bool(a.merge_in_launch), chain_cu_stride=a.chain_cu_stride, chain_sides=a.chain_sides,
                               host_prep=a.host_prep)
    else:
        wl = Workload(S, H, W, B, dev, dynamic, seed0=rank * 1000, multi_view=not a.single_view_launches, pose_seed=114514 + rank, moving_object=chain)
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(a.warmup):
        wl.step(False, order)
    wl.finish()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_s0 = getattr(wl, "host_seconds", 0.0)                      # (the warm-up steps prepared their pairs too)
    st = pipeline.empty_stats()
    for _ in range(a.steps):
        st["pairs"] += wl.step(True, order)
    wl.finish()                                                      # pipeline epilogue: the last pair's Stage B + merge
    # end-of-batch statistics: the ONE collective of the path (SUM / MAX all-reduce of a 7-float vector, RCCL over xGMI)
    total_pairs = int(pipeline.reduce_stats(st)["pairs"])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    my_pairs = st["pairs"]
    host_s_timed = getattr(wl, "host_seconds", 0.0) - host_s0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    else:
        total_pairs = my_pairs

    # the same pairs with the host work of a pair placed differently (outside the timed region; same box, seconds apart): what putting the per-pair pose
    # draws + homographies inside the timed region costs.  "once" = rounds 1-5's headline (fixed poses per image, prepared at set-up)
    host_prep_cmp = {}
    if pipelined and not a.no_sub:
        keep = wl.host_prep
        for mode in ("once", "per-pair", "window"):
            wl.host_prep = mode
            wl.step(False, order[:16])
            wl.finish()
            torch.cuda.synchronize()
            barrier()
            h0, tc0, e0 = wl.host_seconds, time.perf_counter(), len(wl.ev)
            nc = 0
            for _ in range(max(1, min(a.steps, 5))):
                nc += wl.step(True, order)
            wl.finish()
            torch.cuda.synchronize()
            tc = time.perf_counter() - tc0
            lm = float(np.mean([x.elapsed_time(y) for x, y in wl.ev[e0:]])) if len(wl.ev) > e0 else None
            del wl.ev[e0:]                                           # the headline's roofline entry keeps the timed region's launches only
            host_prep_cmp[mode] = {"pairs_per_s": nc / tc, "us_per_pair": tc / nc * 1e6, "pair_launch_us": None if lm is None else lm * 1e3,
                                   "host_prep_us_per_pair": (wl.host_seconds - h0) / nc * 1e6, "pairs": nc}
        wl.host_prep = keep

    # BASELINE configs[3] beside the weak-scaling `value`, on every run: a FIXED batch of --batch images (512) sharded i % world over the ranks,
    # each rank renders its share (cycling through its resident stacks); pairs/s = batch / slowest rank, per-rank seconds reported
    batch_rec = None
    if a.mode == "resident" and a.batch > 0 and dynamic:
        mine_b = pipeline.shard_indices(a.batch, rank, world)
        order_b = list(range(len(mine_b)))
        wl.step(False, order_b[:8])
        wl.finish()
        torch.cuda.synchronize()
        barrier()
        tb0 = time.perf_counter()
        nb = wl.step(False, order_b)
        wl.finish()
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb0
        per_rank = [(tb, nb)]
        if world > 1:
            import torch.distributed as dist
            box = [None] * world
            dist.all_gather_object(box, (tb, nb), group=pipeline.host_side_group())
            per_rank = box
        batch_rec = {"workload": "BASELINE configs[3]: a fixed batch of %d images (64 planes, 640x960, full dynamic pipeline) sharded i %% world over %d rank(s), "
                                 "one pass; strong scaling" % (a.batch, world),
                     "batch_images": a.batch, "pairs_per_s": sum(n for _, n in per_rank) / max(t for t, _ in per_rank), "scaling": "strong",
                     "per_rank_seconds": [t for t, _ in per_rank], "per_rank_pairs": [n for _, n in per_rank]}

    if rank == 0:
        roofs = wl.rooflines()
        traffic, traffic_src = measured_traffic("pair" if pipelined else "stage_b")
        roof = dict(roofs["pair"] if pipelined else roofs["stage_b"])
        roof["traffic"] = traffic
        if traffic_src:
            roof["traffic_source"] = traffic_src
        how = (("pipelined: per pair ONE heterogeneous-grid launch (Stage B of this pair, 2 views + Stage A+C of the next pair, which also merges the pair before "
                "this one as a per-pixel prologue)" if a.merge_in_launch else
                "pipelined: per pair one heterogeneous-grid launch (Stage B of this pair, 2 views + Stage A+C of the next pair) + merge") if pipelined
               else "one kernel after the other: blend + 2 flows, 2 warped views in one launch, merge")
        mo = (" + the moving-object chain of every pair (depth->flow projection, order-preserving forward warp of the uint8 source frame, masks on "
              "disp = rand(H, W): SURVEY 8(d)'s full c3)" + (" on a side stream underneath the next pair launch" if pipelined else "")) if chain else " (render only, no moving-object chain)"
        cfg_name = ("BASELINE configs[2]: %d planes, %dx%d, full dynamic pair (" + how + ")" + mo if dynamic
                    else "BASELINE configs[1]: %d planes, %dx%d, camera-only novel view (blend+flow, warp+composite, u8 frames)") % (S, H, W)
        out = {
            "metric": "image-pairs/sec (+flow) at 640x960x64 planes",
            "value": total_pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if a.mode == "batch" else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg_name, "mode": a.mode, "pipeline": "overlapped" if pipelined else "serial", "moving_object_chain": bool(chain), "merge_in_launch": bool(a.merge_in_launch) if pipelined else None, "chain_cu_stride": a.chain_cu_stride, "chain_side_streams": a.chain_sides, "main_stream_priority": a.main_priority, "chain_ordered_on_main_stream": bool(a.chain_ordered) if chain and pipelined else None,
                       "chain_join": (("per pair on the main stream (event wait before the pair is handed back)" if a.chain_ordered else
                                       "the chain is an independent side pipeline: each pair's moving-object results carry their own `ready` event and are NOT joined "
                                       "with the pair's render results per pair inside the timed region (no consumer runs in this bench); the side stream is joined once, "
                                       "in the timed region's final flush.  --chain-ordered 1 measures the per-pair join (round 4: +12 us per pair)") if chain and pipelined else None),
                       "tune": a.tune, "library": "libmpiflow_hip_witness.so (NOT the product build)" if a.witness else "libmpiflow_hip.so",
                       "host_prep": ({"window": "per pair, timed: every timed pair draws its own two poses (utils/utils.py:207-208) and gets its own per-plane homographies + fp64 "
                                                "inverses (homography_sampler.py:105-122) INSIDE the timed region; the arithmetic is batched over the in-flight window of "
                                                "%d pairs (one batched evaluation, one pinned upload per window - pipeline.prepare_many, bit-identical to per-pair prepare())" % B,
                                      "per-pair": "per pair, timed: one pose draw + prepare() call per timed pair inside the timed region, nothing batched",
                                      "once": "once per image at set-up, OUTSIDE the timed region (fixed poses; rounds 1-5)"}[a.host_prep] if pipelined else "once per image at set-up (serial pipeline)"),
                       "host_prep_us_per_pair_timed": (host_s_timed / max(1, my_pairs) * 1e6) if pipelined else None,
                       "host_prep_comparison": host_prep_cmp or None,
                       "pairs_per_step_per_gpu": len(order), "resident_stacks_per_gpu": B, "timed_seconds": dt,
                       "sharding": "independent images per rank (i % world == rank), stats all-reduce only",
                       "device": _lib.device_info(local),
                       "world_size": world, "backend": ("%s (RCCL over xGMI)" % backend if backend == "nccl" else backend) if world > 1 else "none (single rank)",
                       "launcher": "self (bench.py --gpus N spawned torch.distributed.run)" if os.environ.get("MPIFLOW_SELF_LAUNCHED") else
                                   ("external (torchrun)" if world > 1 else "none"),
                       "ranks": rank_devices},
            "roofline": roof,
        }
        if batch_rec is not None:
            out["batch512"] = batch_rec
        if world > 1 and backend == "nccl":
            try:
                out["config"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:                                        # noqa: BLE001
                out["config"]["rccl_version"] = None
        if not pipelined:
            out["roofline_stage_ac"] = roofs["stage_ac"]
        if a.mode == "batch":
            out["config"]["batch_images"] = a.batch
            out["config"]["pairs_rank0_per_step"] = len(order)
        if getattr(wl, "r", None) is not None and hasattr(wl.r, "close"):
            wl.r.close()                                     # a CU-masked side stream (--chain-cu-stride)
        del wl
        torch.cuda.empty_cache()
        if world == 1 and not a.no_sub and a.mode == "resident":
            sub = []
            c3 = sub_record("c3 serial: BASELINE configs[2], 64x640x960 dynamic pair, one kernel after the other (per-kernel roofline entries)", 64, 640, 960, 4, dev, True, 5)
            sub.append(c3)
            if pipelined:
                out["roofline_stage_b"], out["roofline_stage_ac"] = c3["stage_b"], c3["stage_ac"]
            else:
                sub.append(sub_record("c3 pipelined: Stage B of pair i + Stage A+C of pair i+1 per launch", 64, 640, 960, 4, dev, True, 5, pipelined=True))
            sub.append(sub_record("c3 + moving-object chain (SURVEY 8(d)'s full c3), serial: pair kernels one after the other + the chain's 3 launches on the same stream",
                                  64, 640, 960, 4, dev, True, 5, moving_object=True))
            # the same pairs, same host work per pair (--host-prep), WITHOUT the moving-object chain: `value` / this = what the chain costs the pair rate
            alone = sub_record("c3 render only, pipelined (no moving-object chain: the `value` of rounds 1-3), host prep: %s" % a.host_prep, 64, 640, 960, 8, dev, True, 26,
                               pipelined=True, host_prep=a.host_prep)
            sub.append(alone)
            if pipelined and chain:
                # the same launch WITHOUT the chain's kernels running underneath it: what the kernel does on its own on this box (the headline
                # `roofline` is measured over the timed region, where the chain takes 20 - 26 us per pair from it)
                out["roofline_pair_alone"] = dict(alone["pair"], pairs_per_s=alone["pairs_per_s"])
            if not (pipelined and chain):
                sub.append(sub_record("c3 pipelined + moving-object chain on the side stream (SURVEY 8(d)'s full c3 in the throughput form)", 64, 640, 960, 4, dev, True, 5,
                                      pipelined=True, moving_object=True))
            sub.append(sub_record("c2: BASELINE configs[1], 64x640x960 camera-only pair", 64, 640, 960, 4, dev, False, 5))
            sub.append(sub_record("c1: BASELINE configs[0] shape, 32x384x512 dynamic pair, pipelined (on the GPU: the product has no CPU path)", 32, 384, 512, 8, dev, True, 10, pipelined=True))
            sub.append(sub_record("c1 serial", 32, 384, 512, 8, dev, True, 10))
            sub.append(sub_record("c5: BASELINE configs[4] shape, 128x1024x1536 dynamic pair, random poses, pipelined (parity at this shape: every pixel against the "
                                  "pinned oracle; the golden recorded from the reference itself is 128x512x768 - it cannot allocate this shape in the build container)",
                                  128, 1024, 1536, 2, dev, True, 5, pipelined=True))
            sub.append(sub_record("c5 serial", 128, 1024, 1536, 2, dev, True, 5))
            out["sub"] = sub
            if dynamic:
                out["overlap"] = overlap_record(S, H, W, dev)
        if world == 1 and not a.no_sub:
            out["hbm_reference"] = hbm_reference(dev)
        if world == 1 and not a.no_sub and not a.no_generator and a.mode == "resident":
            try:
                out["roofline_n1"] = n1_record(dev)
            except Exception as e:                                   # noqa: BLE001
                out["roofline_n1"] = {"error": repr(e)}
            try:
                out["generator"] = generator_record()
            except Exception as e:                                   # noqa: BLE001 - a side record must never cost the headline line
                out["generator"] = {"error": repr(e)}
            try:
                # VERDICT r5 item 5 - what ONE GPU can show of an 8-rank node: the same generator confined to 1/8 of this host's logical CPUs with the writer
                # count an 8-rank run gives each rank (cores / (4 x 8)).  Steady state within 5 % of the unrestricted record = a rank keeps its rate on its share
                # of the host; else `writer_stages` names the stage that falls behind and its per-call time IS the 8-GPU generator ceiling
                ncpu = len(os.sched_getaffinity(0))
                share, w8 = max(2, ncpu // 8), max(2, min(32, ncpu // 8 - 6))        # gen_3dphoto_dynamic.default_writers(8)
                hs = generator_record(n_images=320, n_distinct=64, cpus=share, writers=w8)      # the unrestricted record's set: the end-of-run drain weighs the same in both
                hs["host_share"] = {"cpus_of_the_process": share, "cpus_of_the_box": ncpu, "writers": w8,
                                    "note": "decode, submitting and writer threads of the rank all confined to %d of %d logical CPUs (sched_setaffinity); "
                                            "--writers %d = the CLI's default for 8 ranks on this node (share - 6; round 5's cores / 32 = %d writers gave 0.82 of the unrestricted rate, 94 %% busy)" % (share, ncpu, w8, max(2, ncpu // 32))}
                g = out.get("generator", {})
                if isinstance(hs.get("pairs_per_s_steady_state"), float) and isinstance(g.get("pairs_per_s_steady_state"), float):
                    hs["steady_state_vs_unrestricted"] = hs["pairs_per_s_steady_state"] / g["pairs_per_s_steady_state"]
                out["generator_host_share_1of8"] = hs
            except Exception as e:                                   # noqa: BLE001
                out["generator_host_share_1of8"] = {"error": repr(e)}
            try:                                                     # the same generator with the PARITY-GRADE producer (every convolution in fp32 on mpf_pconv): a short run
                out["generator_precise"] = generator_record(n_images=24, n_distinct=24, model_dtype="fp32")
                out["generator_precise"]["producer_precision"] = ("parity-grade engine (--model-dtype fp32): fp32 tensors, products from bf16 pieces on the matrix cores, fp32 blocks of "
                                                                  "64 products carried in fp64 (tests/test_precise_engine.py: closer to the fp64 mirror than torch's own fp32); "
                                                                  "~37 ms per image, so the generator is bound by it")
            except Exception as e:                                   # noqa: BLE001
                out["generator_precise"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            torch.set_num_threads(host_threads)
            out["cpu_baseline"] = cpu_baseline(S, H, W, a.cpu_pairs, chain=chain)
            out["cpu_baseline"]["reference_measured_in_build_container"] = \
                "reference render_3dphoto_dynamic (the same full dynamic pair) 64x640x960: 104.8 s on 8 threads (tests/golden/make_golden.py)"
        # the scalars that matter, inside `config` (the driver's record keeps config / roofline values but only the NAMES of the other sub-records)
        c = out["config"]
        def _get(d, *path):                                          # noqa: E306
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d if isinstance(d, (int, float)) else None
        c["pair_alone_pairs_per_s"] = _get(out, "roofline_pair_alone", "pairs_per_s")
        c["value_over_pair_alone"] = (out["value"] / c["pair_alone_pairs_per_s"]) if c["pair_alone_pairs_per_s"] else None
        c["pair_alone_launch_ms"] = _get(out, "roofline_pair_alone", "avg_launch_ms")
        c["pairs_per_s_host_prep_once"] = _get(c, "host_prep_comparison", "once", "pairs_per_s")
        c["pairs_per_s_host_prep_per_pair"] = _get(c, "host_prep_comparison", "per-pair", "pairs_per_s")
        c["pairs_per_s_host_prep_window"] = _get(c, "host_prep_comparison", "window", "pairs_per_s")
        c["stage_b_2views_frac"] = _get(out, "roofline_stage_b", "frac")
        c["stage_ac_frac"] = _get(out, "roofline_stage_ac", "frac")
        c["batch512_pairs_per_s"] = _get(out, "batch512", "pairs_per_s")
        c["n1_ms_per_image"] = _get(out, "roofline_n1", "ms_per_image")
        c["n1_hbm_frac"] = _get(out, "roofline_n1", "hbm", "frac")
        c["n1_mfma_frac"] = _get(out, "roofline_n1", "mfma", "frac")
        c["n1_precise_fp32_ms"] = _get(out, "roofline_n1", "precise", "fp32", "ms_per_image")
        c["n1_precise_fp64_ms"] = _get(out, "roofline_n1", "precise", "fp64", "ms_per_image")
        c["generator_pairs_per_s_steady"] = _get(out, "generator", "pairs_per_s_steady_state")
        c["generator_pairs_per_s_whole_process"] = _get(out, "generator", "pairs_per_s_whole_process")
        c["generator_precise_pairs_per_s_steady"] = _get(out, "generator_precise", "pairs_per_s_steady_state")
        c["generator_host_share_1of8_pairs_per_s_steady"] = _get(out, "generator_host_share_1of8", "pairs_per_s_steady_state")
        c["generator_host_share_1of8_vs_unrestricted"] = _get(out, "generator_host_share_1of8", "steady_state_vs_unrestricted")
        c["hbm_read_GBps"] = _get(out, "hbm_reference", "read_GBps")
        c["hbm_copy_GBps"] = _get(out, "hbm_reference", "copy_GBps")
        c["cpu_baseline_pairs_per_s"] = _get(out, "cpu_baseline", "value")
        print(json.dumps(out))
    if masked_main is not None:
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
        _lib.check(_lib.load().mpf_stream_destroy(masked_main), "mpf_stream_destroy")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
