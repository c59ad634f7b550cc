"""Import the *reference* MPI-Flow (read-only at /root/reference) on CPU with stubs.

TEST INFRASTRUCTURE ONLY.  This module exists so `make_golden.py` can run the reference's own
Python (fp32, CPU) in the build container and record input/output vectors under tests/golden/.
Nothing here ships and nothing here is importable on the GPU box (no /root/reference there).

Stubs installed:
  * `cv2`          - absent in this image.  `inpaint` records its arguments and returns the image
                     unchanged (its arithmetic is third-party, parity unpinned); `dilate` is a 3x3
                     grey dilation via scipy; `imwrite`/`merge`/`arrowedLine` are inert.
  * `torchvision`  - absent; only imported for `transforms`/`save_image`, never called on the path.
  * `.cuda()`      - identity (the reference hard-codes .cuda(): utils/utils.py:151-153,187,215,229,
                     utils/mpi/homography_sampler.py:16-17, moving_obj.py:20,37-38,46-60).
  * `ctypes.cdll.LoadLibrary("external/forward_warping/libwarping.so")` - redirected to
                     oracle/_ref/libwarping.so, which oracle/Makefile builds with gcc from the
                     reference's own warping.c where it lies.
"""
import ctypes
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("MPIFLOW_REFERENCE", "/root/reference")
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_WARP_SO = os.path.join(REPO_ROOT, "oracle", "_ref", "libwarping.so")

captured = {}  # last arguments seen by the stubbed third-party calls


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "utils", "mpi"))


def _install_cv2_stub():
    cv2 = types.ModuleType("cv2")
    cv2.INPAINT_NS = 0
    cv2.INPAINT_TELEA = 1

    def inpaint(img, mask, radius, flags):
        captured["inpaint"] = dict(img=np.array(img, copy=True), mask=np.array(mask, copy=True),
                                   radius=radius, flags=flags)
        return np.array(img, copy=True)

    def dilate(img, kernel, iterations=1):
        from scipy.ndimage import grey_dilation
        out = np.asarray(img)
        squeeze = out.ndim == 3 and out.shape[-1] == 1
        if squeeze:
            out = out[..., 0]
        for _ in range(iterations):
            out = grey_dilation(out, footprint=np.asarray(kernel) > 0, mode="constant", cval=0)
        return out  # cv2.dilate drops a trailing singleton channel as well

    def merge(chans):
        return np.concatenate([c if c.ndim == 3 else c[..., None] for c in chans], axis=-1)

    def imwrite(path, img):
        captured["imwrite"] = dict(path=path, img=np.array(img, copy=True))
        return True

    cv2.inpaint, cv2.dilate, cv2.merge, cv2.imwrite = inpaint, dilate, merge, imwrite
    cv2.arrowedLine = lambda *a, **k: None
    cv2.imread = lambda *a, **k: None
    cv2.setNumThreads = lambda *a, **k: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *a, **k: None)
    sys.modules["cv2"] = cv2


def _install_torchvision_stub():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    ut = types.ModuleType("torchvision.utils")
    tr.ToTensor = lambda: (lambda im: torch.from_numpy(np.asarray(im)).permute(2, 0, 1).float() / 255)
    ut.save_image = lambda *a, **k: None
    tv.transforms, tv.utils = tr, ut
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.utils": ut})


_installed = False


def install():
    """Idempotently install the stubs and put the reference on sys.path."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if "cv2" not in sys.modules:
        _install_cv2_stub()
    try:
        import torchvision  # noqa: F401
    except Exception:
        _install_torchvision_stub()
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None

    real_load = ctypes.cdll.LoadLibrary

    def load(name):
        if str(name).endswith("libwarping.so"):
            if not os.path.exists(REF_WARP_SO):
                raise RuntimeError("build oracle/_ref first: make -C oracle ref")
            return ctypes.CDLL(REF_WARP_SO)
        return real_load(name)

    ctypes.cdll.LoadLibrary = load
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _installed = True


def modules():
    """Return the reference modules on the hot path."""
    install()
    import geometry
    import utils.mpi.homography_sampler as homography_sampler
    import utils.mpi.mpi_rendering as mpi_rendering
    import utils.mpi.rendering_utils as rendering_utils
    import utils.utils as ref_utils
    import moving_obj
    return types.SimpleNamespace(geometry=geometry, homography_sampler=homography_sampler,
                                 mpi_rendering=mpi_rendering, rendering_utils=rendering_utils,
                                 utils=ref_utils, moving_obj=moving_obj)
