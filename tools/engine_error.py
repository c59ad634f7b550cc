#!/usr/bin/env python3
"""Error of the HIP producer engine (fp16 storage, fp32 accumulate) against the fp32 torch model with the same random parameters, next to
torch's own fp16 autocast: mean and 99.9th percentile of |sigmoid(rgb)| and |sigma| differences - the numbers the absolute bars of
tests/test_conv_engine.py::test_predictor_engine_matches_fp32_model were set from.  usage: engine_error.py [S H W seed]..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model.engine import HipPredictor

dev = torch.device("cuda:0")
cases = [(8, 128, 256, 5), (3, 256, 128, 6), (64, 384, 1280, 1)]
if len(sys.argv) > 4:
    v = [int(x) for x in sys.argv[1:]]
    cases = [tuple(v[i:i + 4]) for i in range(0, len(v), 4)]


def act(r, c):
    return torch.sigmoid(r[:, :3].float()), torch.relu(r[:, 3].float() * c.float()) + 1e-4


def err(x, ref):
    d = (x - ref).abs().flatten()
    if d.numel() > (1 << 26):
        d = d[torch.randint(0, d.numel(), (1 << 26,), device=d.device)]
    return float(d.mean()), float(d.kthvalue(int(d.numel() * 0.999)).values), float(d.max())


for S, H, W, seed in cases:
    m = MPIPredictor(W, H, S).randomize_(seed).eval().to(dev)
    g = torch.Generator().manual_seed(2)
    img, dsp = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 1, H, W, generator=g).to(dev)
    with torch.no_grad():
        ref_raw, ref_cum, _ = m(img, dsp, raw=True)
        with torch.autocast("cuda", dtype=torch.float16):
            h_raw, h_cum, _ = m(img, dsp, raw=True)
    raw, cum, _ = HipPredictor(m, encoder_dtype=None)(img, dsp)
    print("S=%d %dx%d seed %d  cum max err %.2e" % (S, H, W, seed, float((cum - ref_cum[0]).abs().max())))
    for name, got, half, ref in zip(("sigmoid(rgb)", "sigma"), act(raw, cum), act(h_raw[0], h_cum[0]), act(ref_raw[0], ref_cum[0])):
        e, h = err(got, ref), err(half, ref)
        print("   %-13s engine mean %.3e p99.9 %.3e max %.3e | torch fp16 autocast mean %.3e p99.9 %.3e max %.3e | ref range [%.3g, %.3g]"
              % (name, e[0], e[1], e[2], h[0], h[1], h[2], float(ref.min()), float(ref.max())))
    del m
    torch.cuda.empty_cache()
