#!/usr/bin/env python3
"""The phase-decomposed x2-nearest layers of the fast producer engine (k_conv3x3_up, k_conv3x3_up1: MPF_CONV_LD_NEAREST_PHASE) against the gather form
(MPIFLOW_UP_PHASE=0: k_conv3x3 with the LD_NEAREST_PLANE loader) on random sizes / plane counts / weights.  The two forms sum the same products with the weights of
the upsampled channels pre-summed per phase (fp32 sum, one rounding to fp16), so they agree to fp16-rounding level, not bit for bit - and with random weights that
level is amplified through 25 layers (two fp16-storage evaluations of one random network sit as far from each other as each sits from fp32).  The yardstick is
therefore the fp32 torch model: the phase form's error against it must not exceed the gather form's by more than a quarter (plus a floor).  The phase form run twice - and with upconv(0,1) walking 1 / 2 / 4 planes per workgroup - must be BIT-IDENTICAL (a race in the two tile buffers
of k_conv3x3_up1 or in the chunk loop's barriers would show as a mismatch that comes and goes).  usage: soak_engine_phase.py [n_cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor           # noqa: E402
from mpiflow_amd.model.engine import HipPredictor    # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    S, H, W = rng.choice([1, 2, 3, 4, 8, 12, 16, 32]), 128 * rng.randint(1, 3), 128 * rng.randint(1, 4)      # the bottleneck needs multiples of 128
    if S * H * W > 32 * 384 * 1280:
        S = 4
    m = MPIPredictor(W, H, S).randomize_(rng.randint(0, 99)).eval().to(dev)
    g = torch.Generator(device=dev).manual_seed(case)
    img, dsp = torch.rand((1, 3, H, W), generator=g, device=dev), torch.rand((1, 1, H, W), generator=g, device=dev)

    def run(phase, pw=None):
        os.environ["MPIFLOW_UP_PHASE"] = phase
        os.environ["MPIFLOW_PW"] = "" if pw is None else "up1_0=%d" % pw
        raw, cum, _ = HipPredictor(m)(img, dsp)
        return raw.clone(), cum.clone()

    ref, _ = run("0")
    got, cum = run("1")
    with torch.no_grad():
        r32, c32, _ = m(img, dsp, raw=True)                  # the torch modules in fp32 on the GPU
    r32, c32 = r32[0], c32[0]
    ok = True
    for pw in (1, 2, 4, None):
        if pw is None or S % pw == 0:
            again, _ = run("1", pw)
            ok = ok and torch.equal(got, again)
    act = lambda raw, c: (torch.sigmoid(raw[:, :3]), torch.relu(raw[:, 3] * c) + 1e-4)      # noqa: E731  (model/CPN/decoder.py:166-173)
    (rgb_p, sig_p), (rgb_g, sig_g), (rgb_r, sig_r) = act(got, cum), act(ref, cum), act(r32, c32)
    ep = (float((rgb_p - rgb_r).abs().mean()), float((sig_p - sig_r).abs().mean()))             # phase form against fp32
    eg = (float((rgb_g - rgb_r).abs().mean()), float((sig_g - sig_r).abs().mean()))             # gather form against fp32
    close = all(a <= 1.25 * b + 2e-4 for a, b in zip(ep, eg)) and bool(torch.isfinite(got).all())
    print("case %d S=%d %dx%d: repeat / walking %s; mean error against the fp32 model, sigmoid(rgb) / sigma: phase %.2e / %.2e, gather %.2e / %.2e %s" % (
        case, S, H, W, "bit-identical" if ok else "MISMATCH", ep[0], ep[1], eg[0], eg[1], "" if close else "WORSE"), flush=True)
    bad += 0 if (ok and close) else 1
    del m
    torch.cuda.empty_cache()
os.environ.pop("MPIFLOW_UP_PHASE", None)
os.environ.pop("MPIFLOW_PW", None)
print("soak engine phase: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
