#!/usr/bin/env python3
"""The producer's first layer factorised (default) against the materialised form (MPIFLOW_FMN_FACTOR=0) on random sizes / plane counts / seeds:
feature-mask logits and the network's raw output must agree to fp16-rounding level (the synthesised activation equals the materialised one up
to the order of two roundings).  usage: soak_engine_factor.py [n_cases] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model.engine import HipPredictor

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    S, H, W = rng.choice([1, 3, 8, 20]), 128 * rng.randint(1, 3), 128 * rng.randint(1, 4)          # the bottleneck needs multiples of 128
    m = MPIPredictor(W, H, S).randomize_(rng.randint(0, 99)).eval().to(dev)
    g = torch.Generator(device=dev).manual_seed(case)
    img, dsp = torch.rand((1, 3, H, W), generator=g, device=dev), torch.rand((1, 1, H, W), generator=g, device=dev)
    a, b = HipPredictor(m), HipPredictor(m)
    b.fmn.factor = False
    assert a.fmn.factor
    pd = m.plane_disparities(img)[0].contiguous()
    la, lb = a.fmn.logits(img[0], dsp[0, 0], pd), b.fmn.logits(img[0], dsp[0, 0], pd)
    ra, ca, _ = a(img, dsp)
    rb, cb, _ = b(img, dsp)
    torch.cuda.synchronize()
    e_l = float((la - lb).abs().max()) / max(1.0, float(lb.abs().max()))
    e_c = float((ca - cb).abs().max())
    sa, sb = torch.sigmoid(ra[:, :3]), torch.sigmoid(rb[:, :3])
    e_r = float((sa - sb).abs().mean())
    ok = e_l < 5e-3 and e_c < 5e-3 and e_r < 2e-3 and bool(torch.isfinite(ra).all())
    print("case %d S=%d %dx%d: logits rel %.2e  cum mask %.2e  mean |sigmoid(rgb)| %.2e  %s" % (case, S, H, W, e_l, e_c, e_r, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
    del a, b, m
    torch.cuda.empty_cache()
print("soak engine factor: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
