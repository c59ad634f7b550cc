#!/usr/bin/env python3
"""Host-side cost of the generator's per-image submission (no device sync inside the timed calls): where do the milliseconds the
submitting thread spends per image go?  Wraps the functions render_image() calls and accumulates their wall time over N images at the
CLI's shape (64 x 384 x 1280, repeat 5)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import host_math, ops, pipeline, synth  # noqa: E402

S, H, W, R, N = 64, 384, 1280, 5, int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
torch.set_num_threads(1)
acc = {}


def timed(mod, name, label=None):
    fn = getattr(mod, name)
    label = label or "%s.%s" % (getattr(mod, "__name__", type(mod).__name__).split(".")[-1], name)

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc.setdefault(label, [0.0, 0])
            e[0] += time.perf_counter() - t
            e[1] += 1
    setattr(mod, name, w)


r = pipeline.PairRenderer(S, H, W, dev)
K, disp = synth.intrinsics(H, W), synth.plane_disparities(S)
g = torch.Generator(device="cpu").manual_seed(0)
mpi = torch.rand((S, 4, H, W), generator=g).to(dev)
img = torch.rand((3, H, W), generator=g).to(dev)
om = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev)
dstats = pipeline.DeviceStats(dev)
for mod, name in ((host_math, "homographies_multi"), (host_math, "pack_params"), (host_math, "poses_from_parameters"), (ops, "upload_params"),
                  (ops, "src_blend_flow"), (ops, "warp_composite_views"), (ops, "merge"), (ops, "pair_stats"), (ops, "png_scanlines"),
                  (ops, "blend_flow_params"), (ops, "warp_params")):
    timed(mod, name)
timed(r, "prepare_many", "renderer.prepare_many")
timed(r, "run_pairs", "renderer.run_pairs")
timed(r, "blend", "renderer.blend")
timed(dstats, "add", "dstats.add")
import numpy as np  # noqa: E402
np.random.seed(0)


def image():
    pp = []
    for _ in range(R):
        pp.append(host_math.draw_pose_parameters(0.15, profile="v2"))
        pp.append(host_math.draw_pose_parameters(0.15, base_motions=[0, 0, 0], profile="v2"))
    poses = host_math.poses_from_parameters(pp)
    r.blend(mpi, img, K, disp)
    res = r.run_pairs(mpi, img, K, disp, [om] * R, [(poses[2 * i + 1], poses[2 * i]) for i in range(R)])
    for x in res:
        dstats.add(x["flow_mix"], x["fill_mask"])
        ops.png_scanlines(x["frame_mix"])


for _ in range(3):
    image()
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
for _ in range(N):
    image()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per image: host submission %.3f ms, with the device drained %.3f ms" % (t_host / N * 1e3, t_all / N * 1e3))
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("  %-34s %7.3f ms/image  (%5.1f calls/image, %6.1f us/call)" % (k, t / N * 1e3, n / N, t / n * 1e6))
