#!/usr/bin/env python3
"""Stage B over SEVERAL views of one stack: V launches of mpf_warp_composite vs ONE launch of mpf_warp_composite_views.

For V in --views: checks bit-identity of every output, then times both forms with HIP events on the launch stream
(interleaved rounds, median).  Poses alternate camera / dynamic as in a pair (utils/utils.py:207-236); the stack is the
tail-padded interleaved one.  Reports us per view and the fraction of the 8 TB/s roofline at 16*S*N bytes per view."""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib, host_math, ops, synth  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--views", type=str, default="1,2,4,10")
p.add_argument("--planes", type=int, default=64)
p.add_argument("--height", type=int, default=640)
p.add_argument("--width", type=int, default=960)
p.add_argument("--rounds", type=int, default=5)
p.add_argument("--launches", type=int, default=10)
p.add_argument("--images", type=int, default=3)
p.add_argument("--variants", type=str, default="1,20", help="mpf_tune stage_b variants (1 = gather kernel, 20 = LDS-staged kernel)")
a = p.parse_args()

lib = _lib.select_witness()          # the variant keys this tool switches exist in the witness build only (libmpiflow_hip_witness.so)
if os.environ.get("MPF_VIEW_SHIFT"):
    _lib.check(lib.mpf_tune(b"view_shift", int(os.environ["MPF_VIEW_SHIFT"])))
dev = torch.device("cuda:0")
S, H, W = a.planes, a.height, a.width
g = torch.Generator(device=dev).manual_seed(0)
stacks = []
for i in range(a.images):
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
VMAX = max(int(v) for v in a.views.split(","))
views = []
for v in range(VMAX):
    G = host_math.generate_random_pose(0.15, rng=rng) if v % 2 else host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)
    _, H_st = host_math.homographies(G, k_inv, K, d)
    out = dict(rgb=torch.empty((3, H, W), device=dev), objmask=torch.empty((H, W), device=dev),
               rgb_u8=torch.empty((H, W, 3), dtype=torch.uint8, device=dev))
    views.append(dict(dparams=ops.upload_params(ops.warp_params(H_st, k_inv, G, d), dev), quads=quads[v % 2], out=out))


def separate(V, rgba):
    for v in views[:V]:
        ops.warp_composite(rgba, v["quads"], dparams=v["dparams"], out=v["out"], interleaved=2)


def together(V, rgba):
    ops.warp_composite_views(rgba, views[:V], interleaved=2)


def timed(fn, V):
    fn(V, stacks[0])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.launches):
        fn(V, stacks[i % a.images])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.launches * 1e3


alg = 16.0 * S * H * W
res = {}
print("Stage B, %dx%dx%d, V views of one stack: V launches vs one launch (us per VIEW, median of %d rounds x %d)" % (S, H, W, a.rounds, a.launches))
_lib.check(lib.mpf_tune(b"stage_b", 1))
separate(VMAX, stacks[0])
torch.cuda.synchronize()
want_all = [{k: t.clone() for k, t in v["out"].items()} for v in views[:VMAX]]       # variant 1, one launch per view = the pinned kernel
for variant, V in [(int(x), int(v)) for x in a.variants.split(",") for v in a.views.split(",")]:
    _lib.check(lib.mpf_tune(b"stage_b", variant))
    want = want_all[:V]
    for v in views[:V]:
        for t in v["out"].values():
            t.zero_()
    together(V, stacks[0])
    torch.cuda.synchronize()
    same = all(torch.equal(views[i]["out"][k].view(torch.uint8), want[i][k].view(torch.uint8)) for i in range(V) for k in want[i])
    ts, tt = [], []
    for _ in range(a.rounds):
        ts.append(timed(separate, V))
        tt.append(timed(together, V))
    ms, mt = float(np.median(ts)) / V, float(np.median(tt)) / V
    res["v%d_V%d" % (variant, V)] = dict(separate_us_per_view=ms, one_launch_us_per_view=mt, bit_identical=bool(same))
    print("variant %2d V=%2d  bit-identical=%s   separate %7.1f us/view (frac %.3f)   one launch %7.1f us/view (frac %.3f)   x%.2f" % (
        variant, V, same, ms, alg / ms / 1e6 / 8.0, mt, alg / mt / 1e6 / 8.0, ms / mt), flush=True)
print(json.dumps(res))
