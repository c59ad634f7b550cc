#!/usr/bin/env python3
"""Second-view reuse experiment: shift the odd view's position in the strip-major tile sequence by k (mpf_tune
"view_shift") and time the two-view Stage B launch of the serial c3 pairs (8 images with their own random poses); +-4 = one tile row (8 px) down /
up inside a strip, 4 * tiles_y = one strip (128 px) to the right.  Results are bit-identical for every k (scheduling only; checked)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mpiflow_amd import _lib  # noqa: E402

lib = _lib.select_witness()          # the variant keys this tool switches exist in the witness build only (libmpiflow_hip_witness.so)
dev = torch.device("cuda:0")
S, H, W, B = 64, 640, 960, 8
w = bench.Workload(S, H, W, B, dev, True, seed0=0)
tiles_y, ntiles = H // 8, (W // 32) * (H // 8)
w.step(False)
torch.cuda.synchronize()
ref = [t.clone() for t in w.mix]
shifts = [0, 4, 8, 16, 32, 64, ntiles - 4, ntiles - 8, ntiles - 16, ntiles - 32, 4 * tiles_y, ntiles - 4 * tiles_y, 0]
print("shift (tile-sequence positions)   Stage B two-view launch, us (mean over %d images x 5 rounds; per image in round-robin)" % B)
for k in shifts:
    _lib.check(lib.mpf_tune(b"view_shift", k))
    w.step(False)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(w.mix, ref))
    w.ev_b.clear()
    for _ in range(5):
        w.step(True)
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in w.ev_b]).reshape(5, B) * 1e3
    print("%6d  bit-identical=%s  mean %.1f   per image %s" % (k if k < ntiles // 2 else k - ntiles, same, t.mean(), " ".join("%.0f" % v for v in t.mean(0))), flush=True)
_lib.check(lib.mpf_tune(b"view_shift", 0))
del w
torch.cuda.empty_cache()
# the same knob inside the heterogeneous-grid launch (pipelined pairs)
pw = bench.PipelinedWorkload(S, H, W, B, dev, seed0=0)
for k in [0, 8, 16, ntiles - 8, 4 * tiles_y, 0, 8]:
    _lib.check(lib.mpf_tune(b"view_shift", k))
    pw.step(False)
    pw.finish()
    torch.cuda.synchronize()
    pw.ev.clear()
    for _ in range(5):
        pw.step(True)
    pw.finish()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in pw.ev]) * 1e3
    print("pair launch, shift %6d: mean %.1f us over %d launches" % (k if k < ntiles // 2 else k - ntiles, t.mean(), len(t)), flush=True)
_lib.check(lib.mpf_tune(b"view_shift", 0))
