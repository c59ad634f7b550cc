#!/usr/bin/env python3
"""Steady-state per-module timing of the AdaMPI forward (forward hooks + HIP events), 64 x 384 x 1280, random weights."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model.adampi import GatedConvBlock, GatedConv, FeatMaskNetwork, ResnetEncoder

dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
dt = {"fp16": torch.float16, "fp32": None}[sys.argv[1] if len(sys.argv) > 1 else "fp32"]
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
ev = collections.defaultdict(list)


def hook(name):
    def pre(mod, inp):
        e = torch.cuda.Event(enable_timing=True); e.record(); mod._e0 = e
    def post(mod, inp, out):
        e = torch.cuda.Event(enable_timing=True); e.record(); ev[name].append((mod._e0, e, tuple(inp[0].shape)))
    return pre, post


for name, mod in m.named_modules():
    if isinstance(mod, (GatedConvBlock, FeatMaskNetwork, ResnetEncoder)) or (isinstance(mod, GatedConv) and "dispconv" in name) or name in (
            "decoder.conv_down1", "decoder.conv_down2", "decoder.conv_up1", "decoder.conv_up2"):
        pre, post = hook(name)
        mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)


def fwd():
    with torch.no_grad(), torch.autocast("cuda", dtype=dt, enabled=dt is not None):
        return m(img, dsp, raw=True)


for _ in range(3):
    fwd()
torch.cuda.synchronize(); ev.clear()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    fwd()
e1.record(); torch.cuda.synchronize()
tot = e0.elapsed_time(e1) / 3
print("forward %.1f ms (%s)" % (tot, sys.argv[1] if len(sys.argv) > 1 else "fp32"))
acc = 0
for name, lst in ev.items():
    t = sum(a.elapsed_time(b) for a, b, _ in lst) / 3
    acc += t
    print("%-62s in %-22s %7.2f ms" % (name[-62:], lst[0][2], t))
print("hooked modules %.1f ms; rest (concat / upsample / masks / pooling) %.1f ms" % (acc, tot - acc))
