#!/usr/bin/env python3
"""Two forwards of the producer engine at 64 x 384 x 1280 (profiling target)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model.engine import HipPredictor
dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
hp = HipPredictor(m)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
for _ in range(2):
    hp(img, dsp)
torch.cuda.synchronize()
