import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd.model import MPIPredictor
dev = torch.device("cuda:0")
S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
disp = m.plane_disparities(img)
f = m.fmn

def logits(image, dispm, pd):
    _, _, h, w = image.shape
    b, s = pd.shape
    x = torch.cat([image.unsqueeze(1).expand(b, s, 3, h, w), dispm.unsqueeze(1).expand(b, s, 1, h, w), pd[:, :, None, None, None].expand(b, s, 1, h, w)], dim=2).reshape(b * s, 5, h, w)
    c1 = f.conv1(x); c2 = f.conv2(c1); c3 = f.conv3(c2); c5 = f.conv5(f.conv4(c3))
    c6 = f.conv6(torch.cat([f.upsample(c5), c3], dim=1)); c7 = f.conv7(torch.cat([f.upsample(c6), c2], dim=1))
    c8 = f.conv8(torch.cat([f.upsample(c7), c1], dim=1))
    return f.conv9(c8).reshape(b, s, h, w)

def run(chunk, dt):
    with torch.no_grad(), torch.autocast("cuda", dtype=dt, enabled=dt is not None):
        outs = [logits(img, dsp, disp[:, i:i + chunk]) for i in range(0, S, chunk)]
        return torch.softmax(torch.cat(outs, dim=1).float(), dim=1)

for dt in (None, torch.float16):
    ref = run(64, dt)
    for chunk in (64, 32, 16, 8, 4, 2):
        for _ in range(2): out = run(chunk, dt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out = run(chunk, dt)
        torch.cuda.synchronize()
        print("fmn %s chunk %2d planes: %.1f ms  (max diff vs unchunked %.2e)" % ("fp32" if dt is None else "fp16", chunk, (time.perf_counter() - t0) / 3 * 1e3, float((out - ref).abs().max())), flush=True)
