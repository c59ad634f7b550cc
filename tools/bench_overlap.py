#!/usr/bin/env python3
"""Dynamic pairs (c3): how much of the HBM-bound Stage A+C can run underneath the issue-bound Stage B?

  serial      bench.Workload: A+C, Stage B (2 views, one launch), merge - one after the other on one stream (round-2 structure)
  streams=N   the same pairs on N HIP streams, own renderers (kernel-level concurrency only)
  fused       pipeline.OverlappedPairRenderer: Stage B of pair i and Stage A+C of pair i+1 in ONE heterogeneous-grid launch
              (mpf_warp_views_and_blend_next), depth = planes of loads in flight per A+C wave
  fused x2    two such pipelines on two streams (each launch's tail filled by the other stream's head)
Prints pairs/s and, for the fused forms, the mean duration of the fused launch (HIP events on its stream)."""
import argparse
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (Workload, make_image)
from mpiflow_amd import _lib, host_math, pipeline, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--planes", type=int, default=64)
ap.add_argument("--height", type=int, default=640)
ap.add_argument("--width", type=int, default=960)
ap.add_argument("--images", type=int, default=4)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--skip-streams", action="store_true")
ap.add_argument("--only-fused", action="store_true", help="fused depth 4, one and two pipelines, nothing else (A/B of library builds via MPIFLOW_HIP_LIB)")
ap.add_argument("--depth-ab", action="store_true", help="alternate 4 and 8 planes in flight three times (same-box A/B of the A+C role's load depth)")
ap.add_argument("--pmc", action="store_true", help="counter runs: serial single stream + fused depth 4 only")
a = ap.parse_args()
dev = torch.device("cuda:0")
S, H, W, B = a.planes, a.height, a.width, a.images
lib = _lib.select_witness()          # the variant keys this tool switches exist in the witness build only (libmpiflow_hip_witness.so)
if os.environ.get("MPF_VIEW_SHIFT"):                      # experiment: odd views walk the tile sequence this many positions ahead (tools/bench_view_shift.py)
    _lib.check(lib.mpf_tune(b"view_shift", int(os.environ["MPF_VIEW_SHIFT"])))


def serial(n_streams):
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    wls = []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            wls.append(bench.Workload(S, H, W, B, dev, True, seed0=10 * k))
    torch.cuda.synchronize()

    def run(steps):
        for _ in range(steps):
            for i in range(B):
                for wl, st in zip(wls, streams):
                    with torch.cuda.stream(st):
                        wl.pair(i, n_streams == 1 and steps > 2)
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = a.steps * B * n_streams
    extra = ""
    if n_streams == 1:
        r = wls[0].rooflines()
        extra = "  A+C %.1f us  Stage B %.1f us" % (r["stage_ac"]["avg_launch_ms"] * 1e3, r["stage_b"]["avg_launch_ms"] * 1e3)
    print("serial, %d stream(s): %.0f pairs/s (%.1f us per pair)%s" % (n_streams, n / dt, dt / n * 1e6, extra), flush=True)
    del wls
    torch.cuda.empty_cache()


class Fused:
    def __init__(self, seed0, stream):
        self.stream = stream
        with torch.cuda.stream(stream):
            self.r = pipeline.OverlappedPairRenderer(S, H, W, dev)
            K, disp = synth.intrinsics(H, W), synth.plane_disparities(S)
            rng = random.Random(114514 + seed0)
            self.images, self.preps = [], []
            for i in range(B):
                self.images.append(bench.make_image(S, H, W, dev, seed=seed0 + i))
                G_dyn = host_math.generate_random_pose(0.15, rng=rng)
                G_cam = host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)
                self.preps.append(self.r.prepare(K, disp, [G_cam, G_dyn]))
            self.om = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev)
            self.out = (torch.empty((H, W, 2), device=dev), torch.empty((H, W, 3), dtype=torch.uint8, device=dev), torch.empty((H, W), dtype=torch.uint8, device=dev))
        self.ev = []

        def hook(launch):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            self.ev.append((e0, e1))
        self.r.on_fused = hook

    def push(self, i):
        with torch.cuda.stream(self.stream):
            mpi, img = self.images[i]
            self.r.push(mpi, img, self.preps[i], self.om, out=self.out)

    def flush(self):
        with torch.cuda.stream(self.stream):
            self.r.flush()


def fused(n_pipes, depth, ablate=0):
    _lib.check(lib.mpf_tune(b"ovl_depth", depth))
    _lib.check(lib.mpf_tune(b"ovl_ablate", ablate))
    pipes = [Fused(100 * k, torch.cuda.Stream(dev)) for k in range(n_pipes)]
    torch.cuda.synchronize()

    def run(steps):
        for _ in range(steps):
            for i in range(B):
                for p in pipes:
                    p.push(i)
        for p in pipes:
            p.flush()
    run(2)
    torch.cuda.synchronize()
    for p in pipes:
        p.ev.clear()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = a.steps * B * n_pipes
    t_f = sum(e0.elapsed_time(e1) for p in pipes for e0, e1 in p.ev) / max(1, sum(len(p.ev) for p in pipes))
    print("fused%s, depth %d, %d pipeline(s): %.0f pairs/s (%.1f us per pair)  fused launch %.1f us" % (
        {0: "", 1: " [ablation: Stage B workgroups only]", 2: " [ablation: Stage A+C workgroups only]"}.get(ablate, " [knob %d: prio A+C %d, prio B %d]" % (ablate, (ablate >> 2) & 3, (ablate >> 4) & 3)), depth, n_pipes, n / dt, dt / n * 1e6, t_f * 1e3), flush=True)
    del pipes
    torch.cuda.empty_cache()


if a.depth_ab:
    for _ in range(3):
        for depth in (4, 8):
            fused(1, depth)
    for depth in (4, 8):
        fused(2, depth)
    sys.exit(0)
if a.only_fused:
    fused(1, 4)
    fused(2, 4)
    sys.exit(0)
if a.pmc:
    serial(1)
    fused(1, 4)
    sys.exit(0)
print("hbm reference:", {k: (round(v) if isinstance(v, float) else v) for k, v in bench.hbm_reference(dev).items() if k.endswith("GBps")}, flush=True)
serial(1)
if os.environ.get("MPIFLOW_HIP_LIB"):
    sys.exit(0)
if not a.skip_streams:
    serial(2)
for depth in (8, 4):
    fused(1, depth)
fused(2, 4)
for knob in (4, 12, 16, 48):
    fused(1, 4, ablate=knob)
for depth in (4,):
    fused(1, depth, ablate=1)
    fused(1, depth, ablate=2)
_lib.check(lib.mpf_tune(b"ovl_ablate", 0))
