#!/usr/bin/env python3
"""Host hole filling (mpf_inpaint_host, OpenCV's NS restated) on T concurrent threads: frames per second and ms per frame, on a
384 x 1280 frame with ~28 000 hole pixels shaped like the generator's (disocclusion bands + scattered pixels)."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import ops
H, W = 384, 1280
rs = np.random.RandomState(0)
img = (rs.rand(H, W, 3) * 255).astype(np.uint8)
mask = np.zeros((H, W), np.uint8)
for x0 in range(60, W, 160):
    mask[40:340, x0:x0 + 12] = 1
mask |= (rs.rand(H, W) < 0.01).astype(np.uint8)
print("hole pixels", int(mask.sum()), "host threads available", os.cpu_count())
ops.inpaint_host(img, mask, 3, ops.INPAINT_NS)
t0 = time.perf_counter()
for _ in range(5):
    ops.inpaint_host(img, mask, 3, ops.INPAINT_TELEA)
print("Telea, one thread: %.1f ms per frame" % ((time.perf_counter() - t0) / 5 * 1e3))
for T in (1, 4, 8, 16, 32, 64):
    n_each = 6
    def work():
        for _ in range(n_each):
            ops.inpaint_host(img, mask, 3, ops.INPAINT_NS)
    th = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%2d threads: %6.1f frames/s, %5.1f ms per frame per thread" % (T, T * n_each / dt, dt / n_each * 1e3))
