"""Time and error of the parity-grade producer engine (mpiflow_amd/model/precise.py) at the generator's size, per layer.
usage: python tools/bench_precise.py [fp32|x3|x3-nochunk|x3-notile|fp64|both|all] [S H W]
(fp32: the v_mfma_f32_16x16x4_f32 kernels; x3: fp32 tensors, products from three bf16 pieces on the matrix cores - what --model-dtype fp32 runs;
x3-nochunk: without the chunked LDS-tile kernel of the many-channel layers, x3-notile: without either LDS-tile kernel; "all" also prints every fp32-class mode's error against the fp64 engine, which is the
torch modules in double to 1e-9: tests/test_precise_engine.py)"""
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
outs = {}
import mpiflow_amd.model.precise as P                           # noqa: E402
for name, dt, x3 in (("fp32", torch.float32, False), ("x3", torch.float32, True), ("x3-nochunk", torch.float32, True), ("x3-notile", torch.float32, True), ("fp64", torch.float64, False)):
    if not (which == name or which == "all" or (which == "both" and not x3)):
        continue
    P.X3_TILE = name != "x3-notile"
    P.X3_CHUNK = name not in ("x3-notile", "x3-nochunk")
    pp = PrecisePredictor(m, dtype=dt, x3=x3, keep_dtype=True)
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
    outs[name] = (raw.double(), cum.double())
    del pp, raw, cum
    torch.cuda.empty_cache()
if "fp64" in outs and len(outs) > 1:
    act = lambda rc: (torch.sigmoid(rc[0][:, :3]), torch.relu(rc[0][:, 3] * rc[1]) + 1e-4)          # noqa: E731   model/CPN/decoder.py:166-173
    ref = act(outs["fp64"])
    for name, rc in outs.items():
        if name == "fp64":
            continue
        for what, x, r in zip(("sigmoid(rgb)", "sigma"), act(rc), ref):
            d = (x - r).abs().flatten()
            k = max(1, int(d.numel() * 0.999))
            p999 = float(d[torch.randperm(d.numel(), device=d.device)[:4000000]].kthvalue(min(k, 3996000)).values) if d.numel() > 4000000 else float(d.kthvalue(k).values)
            print("%-5s %-13s vs fp64 engine: mean %.2e  p99.9 %.2e  max %.2e" % (name, what, float(d.mean()), p999, float(d.max())))
