#!/usr/bin/env python3
"""Where the non-convolution time of the HIP producer goes (64 x 384 x 1280)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model import engine as E
dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
hp = E.HipPredictor(m)


def ev(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    pd = m.plane_disparities(img)[0]
    lg = hp.fmn.logits(img[0], dsp[0, 0], pd)
    print("softmax over planes        %.3f ms" % ev(lambda: torch.softmax(lg, dim=0)))
    fm = torch.softmax(lg, dim=0)
    print("cumsum                     %.3f ms" % ev(lambda: torch.cumsum(fm, dim=0)))
    cum = torch.cumsum(fm, dim=0)
    print("context mask (cat, 1-x)    %.3f ms" % ev(lambda: 1 - torch.cat([torch.zeros_like(cum[-1:]), cum[:-1]], dim=0)))
    ctx = 1 - torch.cat([torch.zeros_like(cum[-1:]), cum[:-1]], dim=0)
    def pools():
        for k in (2, 4, 8, 16, 32):
            F.adaptive_avg_pool2d(ctx[None], (H // k, W // k)); F.adaptive_avg_pool2d(fm[None], (H // k, W // k))
    print("10 adaptive_avg_pool2d     %.3f ms" % ev(pools))
    def enc(dt):
        with torch.autocast("cuda", dtype=dt, enabled=dt is not None):
            return m.encoder(img, dsp)
    print("encoder fp16 autocast      %.3f ms" % ev(lambda: enc(torch.float16)))
    print("encoder fp32               %.3f ms" % ev(lambda: enc(None)))
    feats = enc(torch.float16)
    d = m.decoder
    def top():
        with torch.autocast("cuda", dtype=torch.float16):
            return d.conv_up2(d.upsample(d.conv_up1(d.upsample(d.conv_down2(d.downsample(d.conv_down1(d.downsample(feats[-1]))))))))
    print("decoder bottleneck         %.3f ms" % ev(top))
    print("5 NHWC fp16 conversions    %.3f ms" % ev(lambda: [E._nhwc16(f) for f in feats]))
    print("whole engine forward       %.3f ms" % ev(lambda: hp(img, dsp)))
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        hp(img, dsp)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host launch time per forward %.3f ms (then %.3f ms drain)" % ((t1 - t0) / 5 * 1e3, (t2 - t1) * 1e3))
