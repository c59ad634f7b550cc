import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
dt = {"fp16": torch.float16, "fp32": None}[sys.argv[1] if len(sys.argv) > 1 else "fp32"]
for _ in range(3):
    with torch.no_grad(), torch.autocast("cuda", dtype=dt, enabled=dt is not None):
        m(img, dsp, raw=True)
torch.cuda.synchronize()
