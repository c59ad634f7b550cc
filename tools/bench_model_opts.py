#!/usr/bin/env python3
"""Which stock-PyTorch knobs help the AdaMPI forward on MI355X (MIOpen): memory format, autocast dtype, find mode."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor

dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)


def bench(tag, model, dtype, n=3):
    def f():
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
            return model(img, dsp, raw=True)
    for _ in range(2):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    print("%-40s %.1f ms" % (tag, (time.perf_counter() - t0) / n * 1e3), flush=True)


for bm in (False, True):
    torch.backends.cudnn.benchmark = bm
    for cl in (False, True):
        m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
        if cl:
            m = m.to(memory_format=torch.channels_last)
        for dt in (None, torch.float16, torch.bfloat16):
            bench("benchmark=%s channels_last=%s %s" % (bm, cl, "fp32" if dt is None else str(dt).split(".")[-1]), m, dt)
        del m
        torch.cuda.empty_cache()
