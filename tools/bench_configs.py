#!/usr/bin/env python3
"""Throughput of the fused pipeline at the BASELINE.json config shapes, on one GPU (run through gpurun).

  c1  32 x 384 x 512   full dynamic pair
  c2  64 x 640 x 960   camera-only pair (the bench.py workload)
  c3  64 x 640 x 960   full dynamic pair (2 views + merge) + the moving-object chain (projection, forward warp, masks)
  c5 128 x 1024 x 1536 full dynamic pair, random poses
Per-kernel times are HIP-event brackets on the launch stream (median over rounds)."""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib, host_math, ops, pipeline, synth  # noqa: E402

_lib.load()
dev = torch.device("cuda:0")


def timed(fn, rounds=7, inner=3):
    ts = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return float(np.median(ts))


def make(S, H, W, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    mpi = torch.empty((S, 4, H, W), device=dev)
    mpi[:, :3] = torch.rand((S, 3, H, W), generator=g, device=dev)
    mpi[:, 3] = torch.relu(3.0 * torch.randn((S, H, W), generator=g, device=dev) - 4.0) + 1e-4
    img = torch.rand((3, H, W), generator=g, device=dev)
    return mpi, img


def config(name, S, H, W, dynamic):
    mpi, img = make(S, H, W, 1)
    om = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev) if dynamic else torch.ones((H, W), device=dev)
    K, disp = synth.intrinsics(H, W), synth.plane_disparities(S)
    rng = random.Random(114514)
    G_dyn = host_math.generate_random_pose(0.15, rng=rng)
    G_cam = host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)
    r = pipeline.PairRenderer(S, H, W, dev, n_views=2 if dynamic else 1)
    prep = r.prepare(K, disp, [G_cam, G_dyn] if dynamic else [G_dyn])
    N = H * W

    def ac():
        ops.src_blend_flow(mpi, img, out_rgba=r.rgba, out_flows=r.flows[:prep["P"]], dparams=prep["blend"], P=prep["P"], src_u8=r.src_u8,
                           obj_mask=om, quads=r.quads[0], quads_complement=r.quads[1] if dynamic else None)

    def b(v):
        ops.warp_composite(r.rgba, r.quads[1 if (dynamic and v == 1) else 0], dparams=prep["warp"][v], out=r.views[v], interleaved=2)

    def merge():
        ops.merge(r.views[0]["rgb"], r.views[1]["rgb"], r.views[0]["objmask"], r.views[1]["objmask"], r.flows[0], r.flows[1], om)

    def pair():
        ac()
        b(0)
        if dynamic:
            b(1)
            merge()

    s2 = torch.cuda.Stream(device=dev)

    def pair_two_streams():          # both views of a dynamic pair concurrently: plane s is fetched from HBM once, the other view finds it in L2/MALL
        ac()
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        b(0)
        with torch.cuda.stream(s2):
            b(1)
        main.wait_stream(s2)
        merge()

    def image_with_repeats(rep=5):   # the reference's `repeat` loop: blend once per image, then flow-only Stage A+C per pair
        r.blend(mpi, img, K, disp)
        for _ in range(rep):
            r.run(mpi, img, prep, om, complement=(False, True) if dynamic else (False,), reuse_blend=True)
            if dynamic:
                merge()

    ac(); b(0)
    t_rep = timed(image_with_repeats, rounds=5, inner=1)
    print("    blend once + 5 pairs of one image (flow-only A+C): %.1f us per pair" % (t_rep / 5), flush=True)
    t_ac, t_b = timed(ac), timed(lambda: b(0))
    t_pair = timed(pair)
    if dynamic:
        print("    dynamic pair with the two Stage B views on two streams: %.1f us" % timed(pair_two_streams), flush=True)
    line = "%-3s %3dx%4dx%4d %-11s A+C %7.1f us (%.2f TB/s r+w)  B %7.1f us (%.3f of 8 TB/s)  pair %8.1f us  %7.1f pairs/s" % (
        name, S, H, W, "dynamic" if dynamic else "camera-only", t_ac, (32.0 * S * N) / t_ac / 1e6, t_b, 16.0 * S * N / t_b / 1e6 / 8.0, t_pair, 1e6 / t_pair)
    if dynamic:
        line += "  merge %.1f us" % timed(merge)
    print(line, flush=True)
    del mpi, r
    torch.cuda.empty_cache()


def moving_object(H, W):
    rs = np.random.RandomState(32)
    base = synth._upsample(rs.rand(max(H // 16, 2), max(W // 16, 2)), H, W) * 0.3 + 0.05
    inst = np.zeros((H, W), np.float32)
    inst[H // 3: 2 * H // 3, W // 3: 2 * W // 3] = 1.0
    disp = torch.from_numpy((base + 0.5 * inst).astype(np.float32)).to(dev)
    rgb = torch.from_numpy(np.floor(rs.rand(H, W, 3) * 256).astype(np.uint8)).to(dev)
    instd = torch.from_numpy(inst).to(dev)
    K = synth.intrinsics(H, W)
    iK = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    K4 = torch.zeros(1, 4, 4); K4[0, 3, 3] = 1; K4[0, :3, :3] = torch.from_numpy(K)
    T1 = host_math.transformation_from_parameters(torch.zeros(1, 1, 3), torch.zeros(1, 3))
    Ti = host_math.transformation_from_parameters(torch.zeros(1, 1, 3), torch.tensor([[0.07, -0.06, 0.08]]))
    P1, Pi = torch.matmul(K4, T1)[0, :3], torch.matmul(K4, Ti)[0, :3]
    state = {}

    def project():
        state["p"] = ops.moving_object_project(disp, iK, P1, Pi, instd)

    def warp():
        p1, z1, sx, sy, fl = state["p"]
        state["w"] = ops.forward_warp(rgb, sx, sy, z1, H, W)

    def masks():
        ops.warp_masks(state["w"])

    project(); warp(); masks()
    print("moving-object chain %dx%d: projection+select %.1f us, forward warp (sort+resolve) %.1f us, masks %.1f us" % (
        H, W, timed(project), timed(warp), timed(masks)), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "c2", "c3", "c5", "mo"]
    if "c1" in which: config("c1", 32, 384, 512, True)
    if "c2" in which: config("c2", 64, 640, 960, False)
    if "c3" in which: config("c3", 64, 640, 960, True)
    if "c5" in which: config("c5", 128, 1024, 1536, True)
    if "mo" in which: moving_object(640, 960)
