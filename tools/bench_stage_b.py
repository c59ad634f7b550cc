#!/usr/bin/env python3
"""Micro-benchmark of the Stage B (warp + composite) kernel variants on one GPU (run through gpurun).

For every variant selected with mpf_tune("stage_b", v): checks that the outputs are bit-identical to variant 0 (the
straightforward kernel that tests/ pin against the oracle), then times N back-to-back launches with HIP events on the
launch stream, interleaving variants across rounds (within-process A/B)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib, host_math, ops, synth  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--variants", type=str, default="0,1,2,3,4,5,6,7,8")
p.add_argument("--planes", type=int, default=64)
p.add_argument("--height", type=int, default=640)
p.add_argument("--width", type=int, default=960)
p.add_argument("--rounds", type=int, default=5)
p.add_argument("--launches", type=int, default=20)
p.add_argument("--images", type=int, default=4)
p.add_argument("--mask", type=int, default=1)
p.add_argument("--aux", type=int, default=1, help="0 = no depth / tgt_mask outputs (as the fused pipeline runs it)")
p.add_argument("--layout", type=int, default=1, help="1 = interleaved, 2 = interleaved + tail padding, 0 = the reference's channel-planar [S,4,H,W] "
               "(variant 0 = round-1 planar kernel, any other variant = k_warp_composite_planar), 3 = separate rgb [S,3,H,W] / sigma [S,1,H,W] "
               "tensors as render_novel_view_dynamic receives them (mpf_warp_composite_split; variant 0 = torch.cat + round-1 planar kernel)")
p.add_argument("--planar-lds", type=int, default=-1, help="layouts 0 / 3: 1 = LDS-staged footprints (k_warp_composite_planar_lds), 0 = gathers (k_warp_composite_planar), -1 = library default")
a = p.parse_args()

lib = _lib.select_witness()          # the variant keys this tool switches exist in the witness build only (libmpiflow_hip_witness.so)
if a.planar_lds >= 0:
    _lib.check(lib.mpf_tune(b"planar_lds", a.planar_lds))
dev = torch.device("cuda:0")
S, H, W = a.planes, a.height, a.width
g = torch.Generator(device=dev).manual_seed(0)
stacks = []
for i in range(a.images):
    rgba = ops.alloc_rgba_stack(S, H, W, dev)
    rgba.copy_(torch.rand((S, H, W, 4), generator=g, device=dev))
    rgba[..., 3] = torch.relu(3.0 * torch.randn((S, H, W), generator=g, device=dev) - 4.0) + 1e-4
    stacks.append(rgba.permute(0, 3, 1, 2).contiguous() if a.layout in (0, 3) else rgba)
if a.layout == 3:
    stacks = [(st[:, :3].contiguous(), st[:, 3:].contiguous()) for st in stacks]
K = synth.intrinsics(H, W)
k_inv = host_math.k_inverse(K)
d = host_math.plane_depths(synth.plane_disparities(S))
aa, tr = synth.bench_pose()
G = host_math.transformation_from_parameters(torch.tensor([[aa]], dtype=torch.float32), torch.tensor([tr], dtype=torch.float32))[0]
H_ts, H_st = host_math.homographies(G, k_inv, K, d)
dparams = ops.upload_params(ops.warp_params(H_st, k_inv, G, d), dev)
om = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev)
quads = ops.mask_quads(om) if a.mask else None
variants = [int(v) for v in a.variants.split(",")]


def run(v, rgba, out=None):
    _lib.check(lib.mpf_tune(b"stage_b", v))
    if a.layout == 3:
        if v == 0:          # what the mirror did before: assemble [S,4,H,W], then the planar kernel
            return ops.warp_composite(torch.cat(rgba, dim=1).contiguous(), quads, dparams=dparams, out=out, interleaved=0)
        return ops.warp_composite_split(rgba[0], rgba[1], quads, dparams=dparams, out=out)
    return ops.warp_composite(rgba, quads, dparams=dparams, out=out, interleaved=(0 if a.layout == 0 else (a.layout if v > 0 else 1)))


ref = run(0, stacks[0])
torch.cuda.synchronize()
ok = {}
for v in variants:
    r = run(v, stacks[0])
    torch.cuda.synchronize()
    ok[v] = all(torch.equal(r[k].view(torch.int32), ref[k].view(torch.int32)) for k in ("rgb", "depth", "tgt_mask") + (("objmask",) if a.mask else ()))
    if not a.aux:
        r2 = run(v, stacks[0], {k: torch.empty_like(t) for k, t in ref.items() if t is not None and k in ("rgb", "objmask")})
        torch.cuda.synchronize()
        ok[v] = ok[v] and all(torch.equal(r2[k].view(torch.int32), ref[k].view(torch.int32)) for k in r2 if r2[k] is not None and k in ("rgb", "objmask"))
times = {v: [] for v in variants}
out = {k: torch.empty_like(t) for k, t in ref.items() if t is not None}
if not a.aux:
    out.pop("depth"); out.pop("tgt_mask")
for rnd in range(a.rounds):
    for v in variants:
        run(v, stacks[0], out)       # warm
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.launches):
            run(v, stacks[i % a.images], out)
        e1.record()
        torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / a.launches * 1e3)
alg = 16.0 * S * H * W
print("Stage B variants at %dx%dx%d, mask=%d, %d launches x %d rounds (us per launch: median / min; GB/s at median; frac of 8 TB/s)" % (S, H, W, a.mask, a.launches, a.rounds))
for v in variants:
    med, mn = float(np.median(times[v])), float(np.min(times[v]))
    print("variant %2d  bit-identical-to-v0=%s  %8.1f / %8.1f us   %7.1f GB/s  frac %.3f" % (v, ok[v], med, mn, alg / med / 1e3, alg / med / 1e3 / 8000))
print(json.dumps({str(v): float(np.median(times[v])) for v in variants}))
