"""Drop-in for the reference's write_flow.py on the generation path: Middlebury .flo files (writeFlow is what
gen_3dphoto_dynamic_v2.py:120 calls; readFlow is what RAFT's loader and vis_flow.py use)."""
import numpy as np

from . import io_formats


def writeFlow(filename, uv, v=None):
    """write_flow.py:74-103: uv [H,W,2], or u and v as two [H,W] arrays -> float32 202021.25, int32 w, int32 h, interleaved u,v"""
    if v is None:
        uv = np.asarray(uv)
        assert uv.ndim == 3
        assert uv.shape[2] == 2
        flow = uv
    else:
        u, v = np.asarray(uv), np.asarray(v)
        assert u.shape == v.shape
        flow = np.stack([u, v], axis=-1)
    io_formats.write_flo(filename, flow)


def readFlow(fn):
    """write_flow.py:14-33: -> [h,w,2] float32, or None (after printing the reference's message) when the magic is wrong"""
    try:
        return io_formats.read_flo(fn)
    except ValueError:
        print("Magic number incorrect. Invalid .flo file")
        return None
