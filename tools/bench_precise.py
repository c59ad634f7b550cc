"""Time and error of the parity-grade producer engine (mpiflow_amd/model/precise.py) at the generator's size, per layer.
usage: python tools/bench_precise.py [fp32|fp64|both] [S H W]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor                      # noqa: E402
from mpiflow_amd.model.precise import PrecisePredictor          # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
S, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (64, 384, 1280)
dev = torch.device("cuda:0")
m = MPIPredictor(W, H, S).randomize_(1).eval().to(dev)
g = torch.Generator().manual_seed(3)
img, dsp = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 1, H, W, generator=g).to(dev)
for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
    if which not in (name, "both"):
        continue
    pp = PrecisePredictor(m, dtype=dt)
    pp(img, dsp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        raw, cum, _ = pp(img, dsp)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    rows, tot = pp.accounting()
    print("%s: %.1f ms per %dx%dx%d image; %.2f TFLOP (real channels) -> %.1f TFLOP/s; %.1f GB materialised -> %.2f TB/s; peak memory %.1f GB"
          % (name, ms, S, H, W, tot["flops"] / 1e12, tot["flops"] / ms / 1e9, tot["bytes"] / 1e9, tot["bytes"] / ms / 1e9, torch.cuda.max_memory_allocated() / 1e9))
    # per-layer times (events around each launch, one more forward)
    import mpiflow_amd.model.precise as P
    times = {}
    orig = P.PConv.__call__

    def timed(self, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(self, *a, **k)
        e1.record()
        times.setdefault(self.name, []).append((e0, e1))
        return out
    P.PConv.__call__ = timed
    pp(img, dsp)
    torch.cuda.synchronize()
    P.PConv.__call__ = orig
    acc = {r["name"]: r for r in rows}
    for k, ev in times.items():
        t = sum(a.elapsed_time(b) for a, b in ev)
        if t > 0.3:
            print("   %-24s %7.2f ms  %6.1f TFLOP/s  %5.2f TB/s" % (k, t, acc[k]["flops"] / t / 1e9, acc[k]["bytes"] / t / 1e9))
    del pp
    torch.cuda.empty_cache()
