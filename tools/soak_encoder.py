#!/usr/bin/env python3
"""mpf_conv2d_f32 / mpf_maxpool3x3s2_f32 on random shapes against torch in fp64: kernel size 1 / 3 / 7, stride 1 / 2, any padding up to k/2, x2 nearest
up-sampling in front, residual, the three activations, 4 .. 512 input channels, 32 .. 512 output channels, frames from 1 x 1 to 40 x 70 (partial
pixel tiles, both split-K variants); and the whole EncoderEngine against the fp64 torch modules at random multiples of 128.
usage: soak_encoder.py [n_cases] [seed]"""
import os, random, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model import engine as E

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    k = rng.choice([1, 3, 3, 3, 7])
    stride, pad, up = rng.choice([1, 2]), rng.randint(0, k // 2), rng.choice([0, 0, 1])
    cin, cout = rng.choice([4, 8, 16, 64, 128, 256, 512]), 32 * rng.choice([1, 2, 3, 4, 8, 16])
    h, w = rng.randint(1, 40 >> up), rng.randint(1, 70 >> up)
    if (h << up) + 2 * pad < k or (w << up) + 2 * pad < k:
        continue
    act, res = rng.choice([None, "relu", "leaky"]), rng.random() < 0.4
    g = torch.Generator().manual_seed(case)
    conv, bn = torch.nn.Conv2d(cin, cout, k, stride, pad, bias=False), torch.nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5); bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1); bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    x = torch.randn(1, cin, h, w, generator=g)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    with torch.no_grad():
        ref = bn.double()(conv.double()(xin.double()))
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    ref = {None: lambda t: t, "relu": torch.relu, "leaky": lambda t: F.leaky_relu(t, 0.1)}[act](ref)
    conv.float(), bn.float()
    nhwc = lambda t: t[0].permute(1, 2, 0).contiguous().to(dev)       # noqa: E731
    out, out16 = E.Conv2dF32(dev, conv, bn, act=act, up=up, slope=0.1)(nhwc(x), residual=nhwc(r) if res else None, f16=True)
    torch.cuda.synchronize()
    ref_hwc = ref[0].permute(1, 2, 0).float().to(dev)
    err, scale = float((out - ref_hwc).abs().max()), max(float(ref_hwc.abs().max()), 1e-3)
    ok = out.shape == ref_hwc.shape and err <= 3e-5 * scale and torch.equal(out16, out.to(torch.float16))
    # max-pool on the same frame
    mp = E.maxpool3x3s2(nhwc(x)) if cin % 4 == 0 else None
    ok = ok and torch.equal(mp, F.max_pool2d(x.to(dev), 3, 2, 1)[0].permute(1, 2, 0).contiguous())
    print("case %d k=%d s=%d p=%d up=%d %d->%d %dx%d act=%s res=%d: err %.2e of %.2e  %s" % (case, k, stride, pad, up, cin, cout, h, w, act, res, err, scale, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
for case in range(3):
    H, W = 128 * rng.randint(1, 3), 128 * rng.randint(1, 4)
    m = MPIPredictor(W, H, 2).randomize_(rng.randint(0, 99)).eval()
    g = torch.Generator().manual_seed(100 + case)
    img, dsp = torch.rand(1, 3, H, W, generator=g), torch.rand(1, 1, H, W, generator=g)
    md = MPIPredictor(W, H, 2).eval(); md.load_state_dict(m.state_dict()); md = md.double()
    md.encoder.img_mean, md.encoder.img_std = md.encoder.img_mean.double(), md.encoder.img_std.double()
    with torch.no_grad():
        feats = md.encoder(img.double(), dsp.double()); d = md.decoder
        top = d.conv_up2(d.upsample(d.conv_up1(d.upsample(d.conv_down2(d.downsample(d.conv_down1(d.downsample(feats[-1]))))))))
    m = m.to(dev)
    _, _, f32 = E.EncoderEngine(m.encoder, m.decoder, dev).forward(img[0].to(dev), dsp[0, 0].to(dev), keep_f32=True)
    torch.cuda.synchronize()
    worst = max(float((got - ref[0].permute(1, 2, 0).float().to(dev)).abs().max()) / float(ref.abs().max()) for got, ref in zip(f32, feats + [top]))
    ok = worst <= 5e-5
    print("encoder %dx%d: worst feature error %.2e of its range  %s" % (H, W, worst, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("soak encoder: %d cases, %d mismatches" % (n_cases + 3, bad))
sys.exit(1 if bad else 0)
