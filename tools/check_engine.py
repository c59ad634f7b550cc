#!/usr/bin/env python3
"""Error budget of the HIP producer engine: engine vs fp32 torch, next to torch fp16-autocast vs fp32 (same parameters)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model.engine import HipPredictor

dev = torch.device("cuda:0")
S, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8, 128, 256)))
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5
m = MPIPredictor(W, H, S).randomize_(seed).eval().to(dev)
g = torch.Generator().manual_seed(2)
img, dsp = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 1, H, W, generator=g).to(dev)


def stats(name, a, b):
    d = (a - b).abs().flatten()
    k = max(1, int(d.numel() * 0.999))
    print("%-34s max %.3e  p99.9 %.3e  mean %.3e   (ref max %.3e)" % (name, float(d.max()), float(d.kthvalue(k).values), float(d.mean()), float(b.abs().max())))


with torch.no_grad():
    r32, c32, _ = m(img, dsp, raw=True)
    with torch.autocast("cuda", dtype=torch.float16):
        r16, c16, _ = m(img, dsp, raw=True)
    raw, cum, _ = HipPredictor(m, encoder_dtype=None)(img, dsp)
    fm32 = m.fmn(img, dsp, m.plane_disparities(img))[0]
    from mpiflow_amd.model.engine import FeatMaskEngine
    fme = FeatMaskEngine(m.fmn, dev)(img[0], dsp[0, 0], m.plane_disparities(img)[0])
r32, c32, r16, c16 = r32[0].float(), c32[0].float(), r16[0].float(), c16[0].float()
stats("feature mask: engine vs fp32", fme, fm32)
for name, (r, c) in (("engine", (raw, cum)), ("torch fp16 autocast", (r16, c16))):
    stats("raw      : %s" % name, r, r32)
    stats("rgb      : %s" % name, torch.sigmoid(r[:, :3]), torch.sigmoid(r32[:, :3]))
    stats("sigma    : %s" % name, torch.relu(r[:, 3] * c) + 1e-4, torch.relu(r32[:, 3] * c32) + 1e-4)
    stats("cum_mask : %s" % name, c, c32)
