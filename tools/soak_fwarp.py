#!/usr/bin/env python3
"""Randomised soak of the forward warp's three paths (gather = default, general radix, round 2's sort + bucket workgroups): random sizes from one pixel up, targets
from smooth flows, white noise, regions clamped onto border pixels (pile-ups of thousands), sentinel / tied / NaN z - every path must write the same
bytes; a subset is checked against the serial C restatement (oracle/).  usage: python tools/soak_fwarp.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib, ops       # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
lib = _lib.select_witness()          # the variant keys this tool switches exist in the witness build only (libmpiflow_hip_witness.so)
try:
    from oracle import mpi_oracle as orc
except Exception:                        # noqa: BLE001
    orc = None
bad = 0
for case in range(cases):
    h = int(rs.choice([1, 2, 3, 7, 16, 33, 64, 100, 257, 384, 640])) if rs.rand() < 0.5 else int(rs.randint(1, 700))
    w = int(rs.choice([1, 2, 5, 17, 64, 65, 300, 511, 960, 1280])) if rs.rand() < 0.5 else int(rs.randint(1, 1300))
    n = h * w
    yy, xx = np.mgrid[0:h, 0:w]
    kind = rs.randint(5)
    if kind == 0:                        # smooth flow (what a depth map gives)
        fx = xx + 8 * np.sin(yy / 37.0) + rs.uniform(-20, 20)
        fy = yy + 5 * np.cos(xx / 53.0) + rs.uniform(-10, 10)
    elif kind == 1:                      # white noise inside a box, identity elsewhere (bench.py's c3)
        fx, fy = xx.astype(np.float64), yy.astype(np.float64)
        m = (yy > h // 4) & (yy < h // 2) & (xx > w // 4) & (xx < w // 2)
        fx = fx + m * rs.uniform(-40, 40, (h, w))
        fy = fy + m * rs.uniform(-25, 25, (h, w))
    elif kind == 2:                      # far out of the frame: whole regions clamp onto the border
        fx = xx + rs.uniform(-2, 2) * w
        fy = yy + rs.uniform(-2, 2) * h
    elif kind == 3:                      # everything onto a few targets
        fx = rs.randint(0, max(1, w // 50), (h, w)).astype(np.float64)
        fy = rs.randint(0, max(1, h // 50), (h, w)).astype(np.float64)
    else:                                # uniformly random targets
        fx, fy = rs.uniform(0, w, (h, w)), rs.uniform(0, h, (h, w))
    idx = np.clip(fx.astype(np.int64), 0, w - 1).reshape(-1)
    idy = np.clip(fy.astype(np.int64), 0, h - 1).reshape(-1)
    z = (rs.randint(0, 16, n) * 0.25 + 0.5).astype(np.float32)
    z[rs.rand(n) < 0.01] = 1000.0
    if rs.rand() < 0.2:
        z[rs.rand(n) < 0.005] = np.nan
    src = rs.randint(0, 256, n * 3).astype(np.uint8)
    t = lambda a: torch.from_numpy(a).to(dev)      # noqa: E731
    outs = []
    for path in (0, 2, 1):
        _lib.check(lib.mpf_tune(b"fwarp_path", path))
        outs.append(ops.forward_warp(t(src), t(idx), t(idy), t(z), h, w).cpu().numpy())
    _lib.check(lib.mpf_tune(b"fwarp_path", 0))
    ok = np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    if ok and orc is not None and n <= 200000 and case % 4 == 0:
        ok = np.array_equal(outs[0].reshape(-1), np.asarray(orc.forward_warping(src, idx, idy, z, h, w)).reshape(-1))
    if not ok:
        bad += 1
        print("MISMATCH case %d: %d x %d kind %d" % (case, h, w, kind), flush=True)
print("soak fwarp: %d cases x 3 paths (+ the serial C restatement on every 4th small case), %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
