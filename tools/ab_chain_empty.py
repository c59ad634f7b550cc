#!/usr/bin/env python3
"""What does the moving-object chain cost the pair stream BESIDES its work?  bench.py's pipelined c3 with the chain's C call replaced by (a) three trivial
kernels on the side stream (a 256-element add each: the launches without the work), (b) only pass 1 (projection + keys), (c) passes 1 + 2, (d) the chain.
usage: python tools/ab_chain_empty.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from mpiflow_amd import moving_obj, ops
from mpiflow_amd import pipeline
mode = os.environ["CHAIN_MODE"]
if mode.endswith("_noguard"):
    mode = mode[:-8]
    _init = pipeline.OverlappedPairRenderer.__init__
    def init(self, *a, **k):
        _init(self, *a, **k)
        self.guard_unconsumed = False
    pipeline.OverlappedPairRenderer.__init__ = init
if mode != "full":
    tiny = {}
    real = ops.moving_object_chain
    def run(self, disp, inst, src_u8, which=None):
        if which is None:
            which, self._next = self._next, (self._next + 1) %% len(self.bufs)
        b = self.bufs[which]
        t = tiny.setdefault(disp.device, torch.zeros(256, device=disp.device))
        if mode == "empty3":
            t.add_(1.0); t.add_(1.0); t.add_(1.0)
        elif mode == "project":
            ops.moving_object_project(disp, self.inv_K, self.P_static, self.P_obj, inst)
        return b
    moving_obj.MovingObjectChain.run = run
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-sub"]
import runpy
runpy.run_path(os.path.join(%r, "bench.py"), run_name="__main__")
''' % (ROOT, ROOT)
for rnd in range(2):
    for mode in ("full", "full_noguard"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CHAIN_MODE=mode), capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("%-8s value %.1f pairs/s, pair launch %.1f us" % (mode, d["value"], d["roofline"]["avg_launch_ms"] * 1e3), flush=True)
        except Exception:                                         # noqa: BLE001
            print(mode, "failed:", (r.stderr or r.stdout)[-600:], flush=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-sub", "--no-moving-object"], capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print("%-8s value %.1f pairs/s, pair launch %.1f us" % ("none", d["value"], d["roofline"]["avg_launch_ms"] * 1e3), flush=True)
