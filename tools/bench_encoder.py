#!/usr/bin/env python3
"""The single-image part of the producer (RGBD ResNet-18 encoder + bottleneck) at 384 x 1280: EncoderEngine (HIP, fp32) against the torch
modules (MIOpen / ATen, fp32), alone and inside the graphed forward; per-convolution times of the HIP path."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model import engine as E

dev = torch.device("cuda:0")
S, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 384, 1280)))
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)


def ev(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


enc = E.EncoderEngine(m.encoder, m.decoder, dev)
dec = E.DecoderEngine(m.decoder, m.encoder.num_ch_enc, dev, amp_dtype=None)
m.encoder.img_mean, m.encoder.img_std = m.encoder.img_mean.to(dev), m.encoder.img_std.to(dev)


def torch_part():
    with torch.no_grad():
        return dec.shared_inputs(m.encoder(img, dsp))


def graphed(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


print("single-image part alone, eager:   HIP %.3f ms   torch %.3f ms" % (ev(lambda: enc(img[0], dsp[0, 0])), ev(torch_part)))
print("single-image part alone, graphed: HIP %.3f ms   torch %.3f ms" % (ev(graphed(lambda: enc(img[0], dsp[0, 0]))), ev(graphed(torch_part))))

orig = E.Conv2dF32.__call__
times = []


def timed(self, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(self, *a, **k); e1.record()
    times.append((self, e0, e1))
    return out


E.Conv2dF32.__call__ = timed
for _ in range(5):
    times.clear()
    enc(img[0], dsp[0, 0])
torch.cuda.synchronize()
E.Conv2dF32.__call__ = orig
tot = 0.0
print("%-28s %5s %5s %3s %2s %2s  %9s %8s %9s" % ("convolution", "Cin", "Cout", "k", "s", "up", "Hout,Wout", "us", "TFLOP/s"))
for L, e0, e1 in times:
    t = e0.elapsed_time(e1)
    tot += t
    c = L.last_call
    print("%-28s %5d %5d %3d %2d %2d  %4d,%4d %8.1f %9.1f" % (L.name, L.cin, L.cout, L.k, L.stride, L.up, c["Hout"], c["Wout"], t * 1e3, L.flops() / t / 1e9))
print("24 convolutions: %.3f ms (event pairs around each launch, eager), %.1f GFLOP" % (tot, sum(L.flops() for L, _, _ in times) / 1e9))

for kind in ("hip", "torch"):
    hp = E.HipPredictor(m, graph=True, encoder=kind)
    print("forward replayed from one hipGraph, %-5s encoder: %.3f ms" % (kind, ev(lambda: hp(img, dsp), n=10)))
