#!/usr/bin/env python3
"""Do two forwards of the fast producer engine overlap?  Two HipPredictor instances (own activation buffers, own hipGraph) replayed on two streams against the
same two forwards one after the other on one stream, 64 x 384 x 1280.  The question behind it: the generator is bound by this network (one image at a time);
if kernels of two images shared the CUs the way Stage B and Stage A+C of neighbouring pairs do, two images in flight would raise its rate.
usage: python tools/bench_engine_two_streams.py [iterations]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor          # noqa: E402
from mpiflow_amd.model import engine as E           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
hp = [E.HipPredictor(m, graph=True) for _ in range(2)]
st = [torch.cuda.Stream(device=dev) for _ in range(2)]
for k in range(2):
    with torch.cuda.stream(st[k]):
        for _ in range(3):
            hp[k](img, dsp)
torch.cuda.synchronize()


def run(two_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(st[k if two_streams else 0]):
            hp[k](img, dsp)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    a, b = run(False), run(True)
    print("one stream %.2f ms per forward | two streams, two images in flight %.2f ms per forward (%+.1f %%)" % (a, b, 100 * (b / a - 1)))
