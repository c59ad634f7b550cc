#!/usr/bin/env python3
"""The walking conv kernels (MpfConvArgs.pw > 1, prefetch on / off) against one plane per workgroup on random sizes / plane counts / weights, several repeats per
case: the whole forward of the fast producer engine and the feature-mask logits must be BIT-IDENTICAL - the walking form only re-schedules (a race in the second
LDS buffers of the prefetch would show as a mismatch that comes and goes).  usage: soak_engine_pw.py [n_cases] [seed] [repeats]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib                         # noqa: E402
from mpiflow_amd.model import MPIPredictor           # noqa: E402
from mpiflow_amd.model.engine import HipPredictor    # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 3
NAMES = ("l2s", "l7", "l8s", "l9", "up0_0", "up1_0", "disp0")
bad = 0
for case in range(n_cases):
    S, H, W = rng.choice([2, 4, 8, 12, 16, 32, 64]), 128 * rng.randint(1, 3), 128 * rng.randint(1, 5)      # the bottleneck needs multiples of 128
    if S * H * W > 64 * 384 * 1280:
        S = 8
    m = MPIPredictor(W, H, S).randomize_(rng.randint(0, 99)).eval().to(dev)
    g = torch.Generator(device=dev).manual_seed(case)
    img, dsp = torch.rand((1, 3, H, W), generator=g, device=dev), torch.rand((1, 1, H, W), generator=g, device=dev)
    pd = m.plane_disparities(img)[0].contiguous()

    def run(pw, pf):
        os.environ["MPIFLOW_PW"] = ",".join("%s=%d" % (n, pw) for n in NAMES)
        _lib.load().mpf_tune(b"conv_pf", pf)
        hp = HipPredictor(m)
        lg = hp.fmn.logits(img[0], dsp[0, 0], pd).clone()
        raw, cum, _ = hp(img, dsp)
        return lg, raw.clone(), cum.clone()

    ref = run(1, 1)
    ok = True
    for pw in [p for p in (2, 4, 8) if S % p == 0]:
        for pf in (1, 0):
            for _ in range(repeats):
                ok = ok and all(torch.equal(a, b) for a, b in zip(ref, run(pw, pf)))
    torch.cuda.synchronize()
    print("case %d S=%d %dx%d: %s" % (case, S, H, W, "bit-identical" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
    del m
    torch.cuda.empty_cache()
_lib.load().mpf_tune(b"conv_pf", 1)
print("soak engine pw: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
