#!/usr/bin/env python3
"""Cold-cache cost of the batch-1 torch encoder + bottleneck (MIOpen) in the producer's first forward: every run gets an empty MIOpen user
cache (MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR in a fresh temp dir), so this is what a fresh box pays.  usage: miopen_cold_start.py [det]"""
import os, sys, tempfile, time
d = tempfile.mkdtemp(prefix="miopen_cold_")
os.environ["MIOPEN_USER_DB_PATH"] = d
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = d
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
dev = torch.device("cuda:0")
m = MPIPredictor(1280, 384, 64).eval().to(dev)
img, dsp = torch.rand(1, 3, 384, 1280, device=dev), torch.rand(1, 1, 384, 1280, device=dev)
torch.backends.cudnn.deterministic = "det" in sys.argv
torch.cuda.synchronize()
for k in range(3):
    t = time.perf_counter()
    with torch.no_grad():
        f = m.encoder(img, dsp)
    torch.cuda.synchronize()
    print("encoder forward %d: %.3f s  (deterministic=%s, MIOPEN_FIND_MODE=%s)" % (k, time.perf_counter() - t, torch.backends.cudnn.deterministic, os.environ.get("MIOPEN_FIND_MODE")))
print("cache dir holds", sum(len(fs) for _, _, fs in os.walk(d)), "files")
