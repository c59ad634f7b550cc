#!/usr/bin/env python3
"""How this ROCm build moves device -> pinned-host copies (the generator's 16 per image): blit kernel (`__amd_rocclr_copyBuffer`, visible in a kernel
trace, occupies CUs) or SDMA engine, and at what rate - under the runtime's knobs (GPU_FORCE_BLIT_COPY_SIZE, HSA_ENABLE_SDMA)."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
for mb in (0.5, 1.5, 4.0):
    n = int(mb * 2 ** 20)
    d = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    for _ in range(5):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    assert torch.equal(h, d.cpu())
    print("D2H %.1f MB: %.1f us per copy, %.1f GB/s   [%s]" % (mb, dt * 1e6, n / dt / 1e9, " ".join("%s=%s" % (k, os.environ[k]) for k in ("GPU_FORCE_BLIT_COPY_SIZE", "HSA_ENABLE_SDMA") if k in os.environ) or "defaults"))
