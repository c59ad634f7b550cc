#!/usr/bin/env python3
"""The submitting thread's WHOLE per-image path in isolation (upload, input stage, network graph replay, blend, source-frame hand-off, instance
masks, five pairs, their hand-off), at the CLI's shape, with the writer jobs stubbed out (a slot returns at once): what does each piece cost the
host when nothing else competes for it?  Wall time per piece (no device sync inside), mean over N images."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import host_math, io_formats, ops, pipeline, synth  # noqa: E402
from mpiflow_amd.model.adampi import MPIPredictor  # noqa: E402
from mpiflow_amd.model.engine import HipPredictor  # noqa: E402

S, H, W, R, N = 64, 384, 1280, 5, int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
torch.set_num_threads(1)
torch.cuda.set_device(dev)
model = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
hip_model = HipPredictor(model, encoder_dtype=None, graph=True)
r = pipeline.PairRenderer(S, H, W, dev)
K = torch.from_numpy(synth.intrinsics(H, W))
ring = io_formats.OutputRing(H, W, dev, slots=64, threads=2, host_fill=lambda f, h: f)
ring._finish = lambda slot, event, *a, **k: ring._free.put(slot)          # writer jobs stubbed: the slot comes straight back
dstats = pipeline.DeviceStats(dev)
rs = np.random.RandomState(0)
item = dict(rgb_u8=torch.from_numpy(rs.randint(0, 255, (375, 1242, 3), dtype=np.uint8)).pin_memory(),
            disp_u8=torch.from_numpy(rs.randint(0, 255, (375, 1242), dtype=np.uint8)).pin_memory(),
            ids_u8=torch.from_numpy((rs.rand(375, 1242) * 3).astype(np.uint8)).pin_memory())
inputs = dict(image=torch.empty((3, H, W), device=dev), disp=torch.empty((H, W), device=dev))
tail_stream = torch.cuda.Stream(device=dev)
acc = {}


class lap:
    def __init__(self, k):
        self.k = k

    def __enter__(self):
        self.t = time.perf_counter()

    def __exit__(self, *e):
        acc.setdefault(self.k, []).append(time.perf_counter() - self.t)


def image():
    with lap("1 upload (3 x .to) + input stage"):
        rgb8, dsp8, ids = (item[k].to(dev, non_blocking=True) for k in ("rgb_u8", "disp_u8", "ids_u8"))
        pre = ops.prepare_inputs(rgb_u8=rgb8, disp_u8=dsp8, size=(H, W), out=inputs)
        image, disp = pre["image"][None], pre["disp"][None, None]
    with lap("2 network: 2 copies + graph replay"):
        mpi, cum_mask, planes = hip_model(image, disp)
    with lap("3 blend"):
        r.blend(mpi, image[0], K, planes, cum_mask=cum_mask)
    with lap("4 source frame: scanlines + hand-off"):
        ring.submit_source(ops.png_scanlines(r.src_u8), ["/dev/null"] * R)
    with lap("5 pose draws + poses_from_parameters"):
        pp = []
        for _ in range(R):
            pp.append(host_math.draw_pose_parameters(0.15, profile="v2"))
            pp.append(host_math.draw_pose_parameters(0.15, base_motions=[0, 0, 0], profile="v2"))
        poses = host_math.poses_from_parameters(pp)
    with lap("6 instance masks (5 x input stage)"):
        oms = [ops.prepare_inputs(ids_u8=ids, obj_index=1 + k % 2, size=(H, W))["mask"] for k in range(R)]
    with lap("7 run_pairs"):
        res = r.run_pairs(mpi, image[0], K, planes, oms, [(poses[2 * i + 1], poses[2 * i]) for i in range(R)], cum_mask=cum_mask)
    ready = torch.cuda.Event()
    ready.record()
    with lap("8 hand-off of 5 pairs (stats, 3 copies, event, job)"):
        with torch.cuda.stream(tail_stream):
            tail_stream.wait_event(ready)
            for x in res:
                for t in (x["frame_mix"], x["fill_mask"], x["flow_mix"]):
                    t.record_stream(tail_stream)
                dstats.add(x["flow_mix"], x["fill_mask"])
                ring.submit_pair_fill(x["flow_mix"], x["frame_mix"], x["fill_mask"], None, [])


def timed(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        with lap(label):
            return fn(*a, **k)
    setattr(obj, name, w)


timed(r, "prepare_many", "7a   prepare_many")
timed(pipeline, "pack_pair_blocks", "7a'    pack_pair_blocks (pinned buffer + fill)")
timed(host_math, "homographies_multi", "7a'    homographies_multi")
calls = []
_sbf = ops.src_blend_flow


def sbf(*a, **k):
    t = time.perf_counter()
    try:
        return _sbf(*a, **k)
    finally:
        calls.append(time.perf_counter() - t)


ops.src_blend_flow = sbf
_up = torch.Tensor.to
timed(ops, "warp_composite_views", "7c   warp_composite_views")
timed(ops, "merge", "7d   merge x 5")
timed(ops, "pair_stats", "8a   pair_stats x 5")
timed(hip_model._graphs if False else torch.cuda.CUDAGraph, "replay", "2a   graph replay")
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
print("per image: host %.3f ms, with the device drained %.3f ms" % (t_host / N * 1e3, t_all / N * 1e3))
print("  piece: mean per image | median-based per image (calls per image x median call: what the piece costs when the launch queue does not push back)")
for k in sorted(acc):
    v = np.array(acc[k])
    print("  %-55s %7.3f ms | %7.3f ms   (max call %.2f ms)" % (k, v.sum() / N * 1e3, len(v) / N * np.median(v) * 1e3, v.max() * 1e3))
c = np.array(calls[-6 * N:]).reshape(N, 6) * 1e6
print("src_blend_flow calls of an image, us (blend, pair 0..4):", " ".join("%.0f" % v for v in c.mean(0)), " median", " ".join("%.0f" % v for v in np.median(c, 0)))
ring.close()
