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
        captured.setdefault("inpaint_calls", []).append(captured["inpaint"])     # "utils/utils copy.py" fills two frames per pair
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


def _torchvision_models_stub():
    """Stand-in for the absent third-party `torchvision.models` - only what model/CPN/encoder.py touches: the `ResNet`
    base class (constructor signature, `_make_layer`, module names conv1/bn1/relu/maxpool/layer1-4/avgpool/fc) and
    `resnet.BasicBlock` / `resnet.Bottleneck`, following torchvision's published layout so that state-dict keys match."""
    import torch.nn as nn
    mod = types.ModuleType("torchvision.models")
    res = types.ModuleType("torchvision.models.resnet")

    class BasicBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, downsample=None):
            super().__init__()
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.downsample = downsample
            self.stride = stride

        def forward(self, x):
            identity = x
            out = self.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            if self.downsample is not None:
                identity = self.downsample(x)
            out += identity
            return self.relu(out)

    class Bottleneck(nn.Module):
        expansion = 4

    class ResNet(nn.Module):
        def __init__(self, block, layers, num_classes=1000):
            super().__init__()
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = self._make_layer(block, 64, layers[0])
            self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
            self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
            self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512 * block.expansion, num_classes)

        def _make_layer(self, block, planes, blocks, stride=1):
            downsample = None
            if stride != 1 or self.inplanes != planes * block.expansion:
                downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                           nn.BatchNorm2d(planes * block.expansion))
            layers = [block(self.inplanes, planes, stride, downsample)]
            self.inplanes = planes * block.expansion
            for _ in range(1, blocks):
                layers.append(block(self.inplanes, planes))
            return nn.Sequential(*layers)

    res.BasicBlock, res.Bottleneck, res.ResNet = BasicBlock, Bottleneck, ResNet
    mod.ResNet, mod.resnet = ResNet, res
    for n in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(mod, n, None)            # only used as dictionary values in ResnetEncoder.__init__
    return mod, res


def _install_torchvision_stub():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    ut = types.ModuleType("torchvision.utils")
    tr.ToTensor = lambda: (lambda im: torch.from_numpy(np.asarray(im)).permute(2, 0, 1).float() / 255)
    ut.save_image = lambda *a, **k: None
    mod, res = _torchvision_models_stub()
    tv.transforms, tv.utils, tv.models = tr, ut, mod
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.utils": ut,
                        "torchvision.models": mod, "torchvision.models.resnet": res})


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


def variant_utils():
    """The two other copies of the per-image module the reference ships: utils/utils_coco.py and
    "utils/utils copy.py" (a file name with a space: loaded by path)."""
    install()
    import importlib.util
    import utils.utils_coco as coco
    spec = importlib.util.spec_from_file_location("utils.utils_copy", os.path.join(REFERENCE_ROOT, "utils", "utils copy.py"))
    copy = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(copy)
    return types.SimpleNamespace(coco=coco, copy=copy)
