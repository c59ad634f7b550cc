#!/usr/bin/env python3
"""Where the generator's start-up goes (bench.py `generator` record: whole process vs steady state): wall seconds of every phase before the
first image of gen_3dphoto_dynamic.py --mpi-from model --model-engine hip at 64 x 384 x 1280, in one process."""
import os, sys, time
t0 = time.perf_counter()
marks = []
def mark(name):
    marks.append((name, time.perf_counter() - t0))
import torch
mark("import torch")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib, io_formats, ops, pipeline
_lib.load()
mark("import mpiflow_amd + dlopen")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
mark("HIP context")
from mpiflow_amd.model import MPIPredictor
m = MPIPredictor(1280, 384, 64)
mark("build MPIPredictor (default init)")
m = m.randomize_(0).eval()
mark("randomize_")
m = m.to(dev); torch.cuda.synchronize()
mark("model.to(device)")
from mpiflow_amd.model.engine import HipPredictor
hp = HipPredictor(m, encoder_dtype=None, graph=True)
torch.cuda.synchronize()
mark("HipPredictor: pack weights of 20 layers + upload")
img, dsp = torch.rand(1, 3, 384, 1280, device=dev), torch.rand(1, 1, 384, 1280, device=dev)
hp(img, dsp); torch.cuda.synchronize()
mark("first forward (MIOpen find, warm-up x2, graph capture, replay)")
hp(img, dsp); torch.cuda.synchronize()
mark("second forward (replay)")
r = pipeline.PairRenderer(64, 384, 1280, dev); torch.cuda.synchronize()
mark("PairRenderer buffers")
ring = io_formats.OutputRing(384, 1280, dev, slots=64, threads=32, host_fill=lambda a, b: a)
mark("OutputRing: 64 pinned slots")
ring.close()
prev = 0.0
for n, t in marks:
    print("%-70s %7.3f s  (+%.3f)" % (n, t, t - prev))
    prev = t
