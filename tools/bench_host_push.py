#!/usr/bin/env python3
"""Host time of the pipelined c3 loop of bench.py: how long does the submitting thread need to ENQUEUE one pair (push + the chain's call + events),
against the ~0.5 ms the GPU needs to render it?  A burst of pairs is enqueued after a sync and timed until the calls return (no sync)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
for chain in (False, True):
    wl = bench.PipelinedWorkload(64, 640, 960, 8, dev, moving_object=chain)
    wl.step(False, list(range(16))); wl.finish(); torch.cuda.synchronize()
    res = []
    for n in (40, 40, 40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.step(False, list(range(n)))
        t1 = time.perf_counter()
        wl.finish(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    print("chain=%s: host enqueue %s us per pair; wall incl. GPU %s us per pair" % (chain, ", ".join("%.0f" % a for a, _ in res), ", ".join("%.0f" % b for _, b in res)))
    del wl
    torch.cuda.empty_cache()
