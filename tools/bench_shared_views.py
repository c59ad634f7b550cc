#!/usr/bin/env python3
"""Verdict r4 item 5 - "one read of the stack for both views": what could ANY sharing scheme gain?

Upper bound by construction: a two-view Stage B launch whose two views use the SAME pose touches every texel twice in the same places, i.e. the
second view's taps are as shareable as taps can be (every plane, not only the far ones).  Timed (HIP events) next to the pair's real two poses
(camera pose + dynamic pose), at view_shift 0 (the two views of a tile dispatched back to back, in lock step) and 8 (the default: de-synchronised);
FETCH_SIZE of the same launches comes from `rocprofv3 --pmc FETCH_SIZE -- python tools/bench_shared_views.py --launches 3` (profiles/r5/).
A scheme that shares only the far planes (the last ~16 of 64, where the two views' source coordinates differ by less than a tile) can collect at
most that share of the difference."""
import argparse
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib, host_math, ops, synth  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--launches", type=int, default=20)
p.add_argument("--rounds", type=int, default=5)
a = p.parse_args()
lib = _lib.select_witness()          # the variant keys this tool switches exist in the witness build only (libmpiflow_hip_witness.so)
dev = torch.device("cuda:0")
S, H, W = 64, 640, 960
g = torch.Generator(device=dev).manual_seed(0)
stacks = []
for i in range(3):
    rgba = ops.alloc_rgba_stack(S, H, W, dev)
    rgba.copy_(torch.rand((S, H, W, 4), generator=g, device=dev))
    rgba[..., 3] = torch.relu(3.0 * torch.randn((S, H, W), generator=g, device=dev) - 4.0) + 1e-4
    stacks.append(rgba)
K = synth.intrinsics(H, W)
k_inv = host_math.k_inverse(K)
d = host_math.plane_depths(synth.plane_disparities(S))
om = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev)
quads = [ops.mask_quads(om, complement=False), ops.mask_quads(om, complement=True)]
rng = random.Random(114514)
G_dyn = host_math.generate_random_pose(0.15, rng=rng)
G_cam = host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)


def view(G, q):
    _, H_st = host_math.homographies(G, k_inv, K, d)
    return dict(dparams=ops.upload_params(ops.warp_params(H_st, k_inv, G, d), dev), quads=q,
                out=dict(rgb=torch.empty((3, H, W), device=dev), objmask=torch.empty((H, W), device=dev)))


cases = {"camera + dynamic pose (the pair)": [view(G_cam, quads[0]), view(G_dyn, quads[1])],
         "camera pose twice (every tap shareable)": [view(G_cam, quads[0]), view(G_cam, quads[1])],
         "dynamic pose twice (every tap shareable)": [view(G_dyn, quads[0]), view(G_dyn, quads[1])],
         "one view, camera pose": [view(G_cam, quads[0])], "one view, dynamic pose": [view(G_dyn, quads[1])]}


def timed(views):
    ops.warp_composite_views(stacks[0], views, interleaved=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.launches):
        ops.warp_composite_views(stacks[i % 3], views, interleaved=2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.launches * 1e3


print("Stage B, 64 x 640 x 960, one launch (us per launch, median of %d x %d)" % (a.rounds, a.launches))
for shift in (8, 0):
    _lib.check(lib.mpf_tune(b"view_shift", shift))
    for name, views in cases.items():
        t = float(np.median([timed(views) for _ in range(a.rounds)]))
        print("view_shift %d  %-42s %7.1f us" % (shift, name, t))
_lib.check(lib.mpf_tune(b"view_shift", 8))
