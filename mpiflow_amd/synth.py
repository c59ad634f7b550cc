"""Deterministic synthetic inputs for the MPI render / flow path.

There is no network for datasets or checkpoints, so tests, goldens and bench.py all use inputs of the shape the
AdaMPI network hands to the path (model/AdaMPI.py:55-78, model/CPN/decoder.py:164-174 of the reference):

  mpi        [S,4,H,W] fp32, channel-planar: rgb in (0,1), sigma = relu(.)+1e-4  (most entries exactly 1e-4)
  disparity  [S]  = linspace(1, 0.001, S+2)[1:-1]   (near -> far; the plane-adjust net is bypassed upstream)
  image      [3,H,W] fp32 in (0,1)
  obj_mask   [H,W]  fp32 in [0,1], soft-edged rectangle (a bilinearly resized instance mask)
  K          [3,3]  = [[.58W,0,.5W],[0,.58H,.5H],[0,0,1]]   (gen_3dphoto_dynamic_v2.py:42-49)

Two value distributions: "white" (i.i.d. per texel - adversarial for parity, |grad sigma| maximal) and "smooth"
(the same draws at 1/8 resolution, bilinearly upsampled - closer to a real network output).
Everything is numpy PCG64 + explicit arithmetic, so the same arrays come out wherever the same numpy runs; goldens
store digests of the inputs they were computed from.
"""
import numpy as np


def intrinsics(H, W):
    return np.array([[0.58 * W, 0.0, 0.5 * W], [0.0, 0.58 * H, 0.5 * H], [0.0, 0.0, 1.0]], dtype=np.float32)


def plane_disparities(S):
    import torch
    return torch.linspace(1.0, 0.001, S + 2)[1:-1].to(torch.float32).numpy().copy()


def _upsample(a, H, W):
    """align_corners=True bilinear upsample of the last two axes, float64, explicit."""
    h, w = a.shape[-2:]
    ys = np.linspace(0.0, h - 1.0, H)
    xs = np.linspace(0.0, w - 1.0, W)
    y0 = np.minimum(np.floor(ys).astype(np.int64), h - 2)
    x0 = np.minimum(np.floor(xs).astype(np.int64), w - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a00 = a[..., y0[:, None], x0[None, :]]
    a01 = a[..., y0[:, None], x0[None, :] + 1]
    a10 = a[..., y0[:, None] + 1, x0[None, :]]
    a11 = a[..., y0[:, None] + 1, x0[None, :] + 1]
    return (a00 * (1 - fx) + a01 * fx) * (1 - fy) + (a10 * (1 - fx) + a11 * fx) * fy


def soft_box_mask(H, W, y0=None, y1=None, x0=None, x1=None, soft=2.0):
    """Rectangle rows [H/4, H/2) x cols [W/4, W/2) with a linear ramp `soft` pixels wide."""
    y0 = H / 4.0 if y0 is None else y0
    y1 = H / 2.0 if y1 is None else y1
    x0 = W / 4.0 if x0 is None else x0
    x1 = W / 2.0 if x1 is None else x1
    yy = np.arange(H, dtype=np.float64)[:, None]
    xx = np.arange(W, dtype=np.float64)[None, :]
    my = np.clip(np.minimum(yy - y0, y1 - yy) / soft + 0.5, 0.0, 1.0)
    mx = np.clip(np.minimum(xx - x0, x1 - xx) / soft + 0.5, 0.0, 1.0)
    return (my * mx).astype(np.float32)


def make_inputs(S, H, W, seed=0, kind="white"):
    """-> dict(mpi [S,4,H,W], disparity [S], image [3,H,W], obj_mask [H,W], K [3,3]) as float32 numpy.

    Drawn plane by plane with numpy's PCG64 Generator straight into float32 (a 64x640x960 stack takes seconds and no
    multi-GB temporaries); goldens record SHA-256 digests of the arrays so a changed stream is detected, not trusted."""
    # "<kind>_opaque": the same draws with the LAST plane made opaque (sigma + 0.05 there: its thickness is 1e3, utils/mpi/mpi_rendering.py:73-78),
    # so the composited weights sum to 1 wherever a ray ends inside the stack and the rendered object masks cluster at 0 / 1 instead of
    # spreading over (0, 1) - and a small object, see below: with a suitable seed NO pixel lies within 1e-5 of the 0.99 threshold (hard part 2) and the
    # thresholded masks compare bit for bit without exclusions (tests/golden/make_golden.py: the *_opaque goldens)
    opaque = kind.endswith("_opaque")
    kind = kind[:-len("_opaque")] if opaque else kind
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    mpi = np.empty((S, 4, H, W), np.float32)
    if kind == "white":
        b1 = np.empty((1, H, W), np.float32)
        for s in range(S):
            g.random(out=mpi[s, :3].reshape(-1), dtype=np.float32)
            g.standard_normal(out=b1, dtype=np.float32)
            mpi[s, 3] = np.maximum(np.float32(3.0) * b1[0] - np.float32(4.0), np.float32(0.0)) + np.float32(1e-4)
        img = g.random((3, H, W), dtype=np.float32)
    elif kind == "smooth":
        h, w = max(H // 8, 2), max(W // 8, 2)
        for s in range(S):
            mpi[s, :3] = _upsample(g.random((3, h, w)), H, W)
            sg = _upsample(g.standard_normal((1, h, w)), H, W)
            mpi[s, 3] = np.maximum(3.0 * sg[0] - 4.0, 0.0) + 1e-4
        img = _upsample(g.random((3, h, w)), H, W).astype(np.float32)
    else:
        raise ValueError(kind)
    om = soft_box_mask(H, W)
    if opaque:
        mpi[-1, 3] += np.float32(0.05)
        # ... and a SMALL object (H/16 x W/16 at the centre): around the object's outline the rendered mask takes every value between 0 and 1
        # over a band as wide as the planes' parallax, so the number of pixels near 0.99 scales with the outline's length
        om = soft_box_mask(H, W, y0=H / 2.0 - H / 32.0, y1=H / 2.0 + H / 32.0, x0=W / 2.0 - W / 32.0, x1=W / 2.0 + W / 32.0)
    return dict(mpi=mpi, disparity=plane_disparities(S), image=img, obj_mask=om, K=intrinsics(H, W))


def bench_pose():
    """Fixed pose for kernel benches: axis-angle (0.01,-0.02,0.005), t = (0.12,-0.08,-0.2)."""
    return (0.01, -0.02, 0.005), (0.12, -0.08, -0.2)
