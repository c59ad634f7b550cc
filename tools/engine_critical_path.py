#!/usr/bin/env python3
"""Which branch of the producer's forward is the critical path before the decoder: the feature-mask network + plane masks on the main stream,
or the batch-1 torch encoder + bottleneck on the side stream (they run concurrently and join in front of the decoder)?  Eager forwards with
events on both streams; times relative to the fork."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model import engine as E

dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
hp = E.HipPredictor(m)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
marks = {}
orig_pm, orig_si = E.plane_masks, E.DecoderEngine.shared_inputs


def pm(*a, **k):
    r = orig_pm(*a, **k)
    e = torch.cuda.Event(enable_timing=True); e.record(); marks["main: feature-mask network + plane masks done"] = e
    return r


def si(self, feats):
    r = orig_si(self, feats)
    e = torch.cuda.Event(enable_timing=True); e.record(); marks["side: encoder + bottleneck done"] = e
    return r


E.plane_masks, E.DecoderEngine.shared_inputs = pm, si
for it in range(4):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hp(img, dsp)
    e1.record()
    torch.cuda.synchronize()
    if it:
        print("forward %.2f ms | %s" % (e0.elapsed_time(e1), " | ".join("%s at %.2f ms" % (k, e0.elapsed_time(v)) for k, v in marks.items())))
