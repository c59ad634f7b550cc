#!/usr/bin/env python3
"""Per-launch timing of the HIP producer engine at 64 x 384 x 1280 (random weights), next to the torch/MIOpen forward."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model import engine as E

dev = torch.device("cuda:0")
S, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 384, 1280)))
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
hp = E.HipPredictor(m)

times = collections.OrderedDict()
orig = E.ConvLayer.__call__


def timed(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(self, *a, **k); e1.record()
    times.setdefault(id(self), []).append((e0, e1, self, a[:3]))
    return out


def ev(fn, n=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_eng = ev(lambda: hp(img, dsp))
E.ConvLayer.__call__ = timed
ITERS = int(os.environ.get("BENCH_ENGINE_ITERS", "3"))
for _ in range(ITERS):
    hp(img, dsp)
torch.cuda.synchronize()
E.ConvLayer.__call__ = orig
tot = 0
print("%-6s %-4s %-3s %-5s %-14s %9s %9s %9s  %s" % ("loader", "epi", "ct", "nblk", "S,Hin,Win", "ms", "TFLOP/s", "GB/s(out)", "layer (blocks per workgroup, weights through LDS)"))
for lst in times.values():
    t = sum(a.elapsed_time(b) for a, b, _, _ in lst) / len(lst)
    L, (s, hin, win) = lst[0][2], lst[0][3]
    hout, wout = (hin - 1) // L.stride + 1, (win - 1) // L.stride + 1
    flops = 2.0 * s * hout * wout * L.nblk * 16 * L.nchunk * E.ksteps(L.ct) * 32
    outb = s * hout * wout * (L.Cst * 2 if L.epi in (0, 2, 6) else 4 * L.Cst)      # fp16 NHWC epilogues (affine, gated, gated interleaved) / fp32 ones
    tot += t
    print("%-6d %-4d %-3d %-5d %-14s %9.3f %9.1f %9.1f  %s (%d, %s)" % (L.loader, L.epi, L.ct, L.nblk, "%d,%d,%d" % (s, hin, win), t, flops / t / 1e9, outb / t / 1e6,
                                                                    L.name, L.nblk // L.ncg, "lds" if E._wlds(L.name, L.wlds_default) else "per wave"))
print("engine forward %.2f ms (conv launches %.2f ms)" % (t_eng, tot))
fm = lambda: hp.fmn(img[0], dsp[0, 0], m.plane_disparities(img)[0])
print("feature-mask network on the engine %.2f ms" % ev(fm))
if "--torch" in sys.argv:
    with torch.no_grad():
        print("torch fp32 forward %.1f ms" % ev(lambda: m(img, dsp, raw=True)))
        with torch.autocast("cuda", dtype=torch.float16):
            print("torch fp16-autocast forward %.1f ms" % ev(lambda: m(img, dsp, raw=True)))
hg = E.HipPredictor(m, graph=True)
hg(img, dsp)
print("engine forward replayed from one hipGraph %.2f ms" % ev(lambda: hg(img, dsp), n=10))
