#!/usr/bin/env python3
"""Does the Infinity Cache (256 MiB, memory side) serve a buffer that a kernel has JUST written?  (Question behind DESIGN.md section 5 / 12: could
Stage B read, from the cache, the part of the blended stack Stage A+C wrote a moment ago?)  For X MB: time a read-only pass over a buffer (a) cold -
after streaming 2 GiB of something else -, (b) right after a read of the same buffer, (c) right after a copy kernel wrote it (streaming stores, as
Stage A+C's), each as GB/s.  mpf_stream_probe's kernels: 16 B per lane, grid-stride."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
flush = torch.empty(2 << 28, dtype=torch.float32, device=dev).normal_()          # 2 GiB
sink = torch.empty(1 << 20, dtype=torch.float32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731


def read(t):
    lib.mpf_stream_probe(p(t), p(sink), t.numel() * 4, 0, st)


def copy(a, b):
    lib.mpf_stream_probe(p(a), p(b), a.numel() * 4, 1, st)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


print("%8s %14s %14s %18s" % ("MB", "cold GB/s", "re-read GB/s", "after-write GB/s"))
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb << 18
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    res = []
    for mode in ("cold", "reread", "written"):
        best = 1e9
        for _ in range(5):
            read(flush)
            if mode == "reread":
                read(b)
            elif mode == "written":
                copy(a, b)
            best = min(best, timed(lambda: read(b)))
        res.append(n * 4 / best / 1e9)
    print("%8d %14.0f %14.0f %18.0f" % (mb, *res), flush=True)
    del a, b
