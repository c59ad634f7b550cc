#!/usr/bin/env python3
"""Five replays of the producer engine's hipGraph at 64 x 384 x 1280 (profiling target: which kernels sit between the convolutions)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model.engine import HipPredictor
dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
hp = HipPredictor(m, graph=True)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
for _ in range(6):
    hp(img, dsp)
torch.cuda.synchronize()
