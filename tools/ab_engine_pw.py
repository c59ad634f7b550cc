#!/usr/bin/env python3
"""Planes per workgroup (MpfConvArgs.pw) of the few-block layers of the fast producer engine: per-layer time and the replayed forward for a list of
MPIFLOW_PW settings, in ONE process on one box, and whether every setting reproduces the pw = 1 outputs bit for bit.
usage: python tools/ab_engine_pw.py ["" "l8s=2" "l8s=4,l9=4" ...]     (64 x 384 x 1280, random weights)"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor          # noqa: E402
from mpiflow_amd.model import engine as E           # noqa: E402

settings = sys.argv[1:] or ["", "l8s=2", "l8s=4"]
dev = torch.device("cuda:0")
S, H, W = (int(v) for v in os.environ.get("AB_SHAPE", "64,384,1280").split(","))
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
orig = E.ConvLayer.__call__
NAMES = ("l2s", "l7", "l8s", "l9", "up0_0", "up1_0", "disp0")


def ev(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
from mpiflow_amd import _lib                        # noqa: E402
for setting in settings:
    os.environ["MPIFLOW_PW"] = setting              # an item "pf=0" / "pf=1" (not a layer name) switches the walking kernels' prefetch: mpf_tune("conv_pf", v)
    pf = [it.split("=")[1] for it in setting.split(",") if it.startswith("pf=")]
    _lib.load().mpf_tune(b"conv_pf", int(pf[0]) if pf else 1)
    bt = [it.split("=")[1] for it in setting.split(",") if it.startswith("bt=")]            # "bt=0": B' of the factorised first layer as a map instead of its border-class table
    os.environ["MPIFLOW_BPRIME_TABLE"] = bt[0] if bt else "1"
    hp = E.HipPredictor(m)
    out = [t.clone() for t in hp(img, dsp)]
    if ref is None:
        ref = out
    same = all(torch.equal(a, b) for a, b in zip(ref, out))
    times = collections.OrderedDict()

    def timed(self, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = orig(self, *a, **k)
        e1.record()
        times.setdefault(self.name, []).append((e0, e1))
        return o

    for _ in range(2):
        hp(img, dsp)
    E.ConvLayer.__call__ = timed
    for _ in range(5):
        hp(img, dsp)
    torch.cuda.synchronize()
    E.ConvLayer.__call__ = orig
    per = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in times.items()}
    hg = E.HipPredictor(m, graph=True)
    hg(img, dsp)
    t_graph = ev(lambda: hg(img, dsp))
    print("MPIFLOW_PW=%-40s bit-equal to the first setting: %s   conv launches %.3f ms   forward (hipGraph) %.3f ms" % (repr(setting), same, sum(per.values()), t_graph))
    print("    " + "  ".join("%s %.3f" % (k, per[k]) for k in NAMES if k in per))
    del hp, hg
