#!/usr/bin/env python3
"""Verdict r4 item 1 (b): "spend the HBM headroom - materialise the synthesised inputs once and stage the consumer's tile with plain copies".
What the consumer would then cost, measured WITHOUT charging the producer anything: the feature-mask network's l8 / l7 / l6 and the decoder's up1_0
with their inputs handed over as ONE plain fp16 NHWC tensor (x2 up-sampling, concatenation and first-layer synthesis done beforehand, untimed) through the
engine's plain loader (LD_DIRECT: one 16-byte load per staged vector, no interpolation, no synthesis), next to the synthesising loaders of the forward.
Lower bound for variant (b): the producer-side write of the materialised tensor (2 - 4 GB per layer) and any extra pass are NOT in these times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor          # noqa: E402
from mpiflow_amd.model import engine as E           # noqa: E402

dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
hp = E.HipPredictor(m)
f = hp.fmn
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
hp(img, dsp)
torch.cuda.synchronize()


def ev(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


A = E.ConvLayer.affine_relu
pd = hp._plane_disp
A1, B1 = f.first_layer_maps(img[0].float().contiguous(), dsp[0, 0].float().contiguous())
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.rand(s, generator=g, device=dev).to(torch.float16)      # noqa: E731
cases = []
# l8: up(c7) 32 ch ++ c1 16 ch at full resolution
c7 = rnd(S, H // 2, W // 2, 32)
cases.append(("l8s (x2 bilinear of c7 + first-layer synthesis in the loader)", lambda: f.l8s(S, H, W, srcA=c7, srcB=A1, cm=B1, plane_vals=pd, HA=H // 2, WA=W // 2),
              A(dev, m.fmn.conv8, [(48, 48)], loader=E.LD_DIRECT, stride=1, ct=16, name="l8m"), (S, H, W, 48)))
c6, c2 = rnd(S, H // 4, W // 4, 64), rnd(S, H // 2, W // 2, 32)
cases.append(("l7 (x2 bilinear of c6 ++ c2)", lambda: f.l7(S, H // 2, W // 2, srcA=c6, srcB=c2, HA=H // 4, WA=W // 4),
              A(dev, m.fmn.conv7, [(96, 96)], loader=E.LD_DIRECT, stride=1, ct=16, name="l7m"), (S, H // 2, W // 2, 96)))
c5, c3 = rnd(S, H // 8, W // 8, 128), rnd(S, H // 4, W // 4, 64)
cases.append(("l6 (x2 bilinear of c5 ++ c3)", lambda: f.l6(S, H // 4, W // 4, srcA=c5, srcB=c3, HA=H // 8, WA=W // 8),
              A(dev, m.fmn.conv6, [(192, 192)], loader=E.LD_DIRECT, stride=1, ct=16, name="l6m"), (S, H // 4, W // 4, 192)))
print("layer: synthesising loader (the forward's) vs plain loader on a materialised input, 64 x 384 x 1280; GB = what the materialised input adds to HBM traffic (written once, read once)")
for name, synth, plain, shp in cases:
    x = rnd(*shp)
    t_s = ev(synth)
    t_p = ev(lambda: plain(shp[0], shp[1], shp[2], srcA=x))
    gb = 2.0 * x.numel() * 2 / 1e9
    print("%-62s %6.3f ms   materialised input, plain loader %6.3f ms  (+ %.1f GB of traffic = + %.2f ms at 5 TB/s, not charged)" % (name, t_s, t_p, gb, gb / 5.0))
    del x
    torch.cuda.empty_cache()
