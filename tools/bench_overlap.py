#!/usr/bin/env python3
"""Dynamic pairs (c3) on 1, 2 or 3 HIP streams: does an HBM-bound Stage A+C of one pair overlap the issue-bound Stage B of another?
Each stream owns its PairRenderer (its own blended stack and outputs) and renders its own images; pairs/s over all streams."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (Workload, make_image)

dev = torch.device("cuda:0")
S, H, W = 64, 640, 960
for n_streams in (1, 2, 3):
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    wls = []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            wls.append(bench.Workload(S, H, W, 4, dev, True, seed0=10 * k))
    torch.cuda.synchronize()

    def run(steps):
        for _ in range(steps):
            for i in range(4):
                for wl, st in zip(wls, streams):
                    with torch.cuda.stream(st):
                        wl.pair(i, False)
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 10
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d stream(s): %.0f pairs/s (%.1f us per pair)" % (n_streams, steps * 4 * n_streams / dt, dt / (steps * 4 * n_streams) * 1e6))
    del wls
    torch.cuda.empty_cache()
