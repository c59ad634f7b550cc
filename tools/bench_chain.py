"""The moving-object chain (mpf_moving_object_chain) ALONE on bench.py's c3 inputs (disp = rand(H, W), the soft box instance mask): time per call of the
default gather path and of round 2's sort path (mpf_tune("fwarp_path", 2)), and that both write the same bytes.
usage: python tools/bench_chain.py [H W] [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from mpiflow_amd import _lib, synth                              # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 960)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
lib = _lib.select_witness()          # the variant keys this tool switches exist in the witness build only (libmpiflow_hip_witness.so)
chain, disp = bench.make_moving_object_chain(H, W, synth.intrinsics(H, W), dev, 0)
inst = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev)
img = torch.rand((3, H, W), device=dev)
ref = None
for path, name in ((0, "gather (round 5)"), (2, "sort + bucket workgroups (round 2)"), (0, "gather (round 5)")):
    _lib.check(lib.mpf_tune(b"fwarp_path", path))
    b = chain.run(disp, inst, img, which=0)
    torch.cuda.synchronize()
    got = [t.clone() for t in (b.warped, b.p1, b.z1, b.safe_x, b.safe_y, b.flow_01)] + [b.masks[k].clone() for k in sorted(b.masks)]
    if ref is None:
        ref = got
    else:
        assert all(torch.equal(x, y) for x, y in zip(got, ref)), "paths differ"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        chain.run(disp, inst, img, which=0)
    e1.record()
    torch.cuda.synchronize()
    print("%-40s %7.1f us per chain (%d x %d, %d calls back to back)" % (name, e0.elapsed_time(e1) / iters * 1e3, H, W, iters))
_lib.check(lib.mpf_tune(b"fwarp_path", 0))
