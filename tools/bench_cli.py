#!/usr/bin/env python3
"""End-to-end generator throughput: gen_3dphoto_dynamic.py on a synthetic KITTI-shaped dataset (375x1242 PNGs -> 384x1280,
64 planes, repeat 5), network on the HIP engine vs torch, with and without writer threads."""
import os, shutil, subprocess, sys, tempfile, time
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_img = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
tmp = tempfile.mkdtemp(prefix="mpf_cli_")
base = os.path.join(tmp, "data")
for d in ("images", "disps", "masks"):
    os.makedirs(os.path.join(base, d))
rs = np.random.RandomState(0)
yy, xx = np.mgrid[0:375, 0:1242]
for i in range(n_img):
    img = (np.clip(0.5 + 0.25 * np.sin(xx / (17.0 + i)) + 0.25 * np.cos(yy / 23.0) + 0.05 * rs.randn(375, 1242), 0, 1) * 255).astype(np.uint8)
    Image.fromarray(np.stack([img, np.roll(img, 7, 1), np.roll(img, 13, 0)], -1)).save(os.path.join(base, "images", "%04d.png" % i))
    Image.fromarray((255 * (0.1 + 0.8 * yy / 375)).astype(np.uint8)).save(os.path.join(base, "disps", "%04d.png" % i))
    m = np.zeros((375, 1242), np.uint8); m[150:300, 300:600] = 1; m[200:330, 800:1000] = 2
    Image.fromarray(m).save(os.path.join(base, "masks", "%04d.png" % i))
configs = [("hip engine, peel fill on the GPU, 16 writers (warm-up run)", ["--model-engine", "hip", "--writers", "16", "--inpaint", "peel"]),
           ("hip engine, peel fill on the GPU, 16 writers", ["--model-engine", "hip", "--writers", "16", "--inpaint", "peel"]),
           ("hip engine, no fill, 16 writers", ["--model-engine", "hip", "--writers", "16", "--inpaint", "none"]),
           ("hip engine, cv2.inpaint NS restated on 8 writer threads", ["--model-engine", "hip", "--writers", "8", "--inpaint", "builtin"]),
           ("hip engine, cv2.inpaint NS restated on 32 writer threads", ["--model-engine", "hip", "--writers", "32", "--inpaint", "builtin"]),
           ("hip engine, cv2.inpaint NS restated on 96 writer threads", ["--model-engine", "hip", "--writers", "96", "--inpaint", "builtin"])]
if "--torch" in sys.argv:
    configs.append(("torch fp16, 16 writers", ["--model-engine", "torch", "--model-dtype", "fp16", "--writers", "16", "--inpaint", "peel"]))
only = os.environ.get("ONLY")                  # substring filter on the configuration labels
for label, extra in [c for c in configs if not only or any(o in c[0] for o in only.split("|"))]:
    out = os.path.join(tmp, "out_" + label.replace(" ", "_").replace(",", ""))
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, os.environ.get("GEN_SCRIPT", "gen_3dphoto_dynamic.py")), "--base", base, "--out", out, "--repeat", "5", "--mpi-from", "model",
                        "--ckpt_path", "random:0"] + extra, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    last = [l for l in r.stdout.splitlines() if l.startswith("pairs")]
    print("%-34s process %.1f s | %s" % (label, dt, last[-1] if last else r.stderr[-400:]))
    for l in r.stdout.splitlines():
        if l.startswith("  ") or l.startswith("steady"):
            print("      " + l)
    shutil.rmtree(out, ignore_errors=True)          # 800 pairs = ~4 GB of files per configuration: do not let them pile up on the box
shutil.rmtree(tmp, ignore_errors=True)
