#!/usr/bin/env python3
"""End-to-end timing with the AdaMPI producer in the loop (random weights): network forward, then the fused render pair."""
import os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import host_math, pipeline, synth
from mpiflow_amd.model import MPIPredictor

dev = torch.device("cuda:0")
# the decoder needs H/32 and W/32 divisible by 4 (two 2x poolings + two 2x upsamplings on the coarsest map): the reference's
# 384 x 1280 default qualifies, 640 x 960 does not
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 384, 1280
for dtype in (None, torch.float16, torch.bfloat16):
    model = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
    img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
    om = torch.from_numpy(synth.soft_box_mask(H, W)).to(dev)
    K, r = synth.intrinsics(H, W), pipeline.PairRenderer(S, H, W, dev)
    rng = random.Random(1)
    Gd, Gc = host_math.generate_random_pose(0.15, rng=rng), host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)

    def net():
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=dtype is not None):
            raw, cm, pd = model(img, dsp, raw=True)
        return raw[0].float().contiguous(), cm[0].float().contiguous(), pd[0].float()

    def run(fn, n=3):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): out = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out

    t_net, (raw, cm, pd) = run(net)
    t_pair, _ = run(lambda: pipeline.render_pair(img[0], om, raw, pd, K, Gc, Gd, renderer=r, cum_mask=cm), n=5)
    print("%dx%dx%d  network forward (%s): %.1f ms   fused dynamic pair (raw hand-off): %.3f ms   peak mem %.1f GB" % (
        S, H, W, "fp32" if dtype is None else str(dtype).split(".")[-1], t_net, t_pair, torch.cuda.max_memory_allocated() / 2**30), flush=True)
    del model, r, raw, cm
    torch.cuda.empty_cache()
