"""Tensor-level wrappers over the C ABI (include/mpiflow_hip.h): torch CUDA tensors in, torch CUDA tensors out.

PyTorch is plumbing here (device memory, current stream); all arithmetic on tensors happens in the HIP kernels of
libmpiflow_hip.so.  Every function launches asynchronously on torch's current stream and raises MpiFlowHipError if
the library is missing or a launch fails - there is no eager/CPU fallback.
"""
import ctypes

import torch

from . import _lib, host_math

_f32 = torch.float32


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(fn):
    """Run `fn` with the device of its first CUDA tensor (or torch.device) argument current, so that the HIP launch and
    torch's "current stream" both refer to the GPU that owns the buffers, whatever device the caller had selected."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                dev = a.device
                break
            if isinstance(a, torch.device) and a.type == "cuda":
                dev = a
                break
        if dev is None or dev.index is None:
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _dev(t, name, dtype=_f32):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.MpiFlowHipError("%s must live on the GPU (got %s); mpiflow_amd has no CPU path" % (name, t.device))
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def upload_params(host_buf, device):
    """Small per-call matrices -> device (one async H2D copy of a few KB on the current stream)."""
    return host_buf.pin_memory().to(device=device, non_blocking=True)


# ---- fused hot path ---------------------------------------------------------------------------------------------

def blend_flow_params(K_inv, depth_S, homs_tgt_src=None):
    """Host image of d_params for mpf_src_blend_flow.  homs_tgt_src: None or [P,S,3,3] (P <= 2).  -> (buf, P)"""
    d = host_math._cpu32(depth_S).reshape(-1)
    S = d.numel()
    if homs_tgt_src is None:
        return host_math.pack_params(K_inv=K_inv, depths=d), 0
    hts = host_math._cpu32(homs_tgt_src).reshape(-1, S, 3, 3)
    P = hts.shape[0]
    homs = hts.permute(1, 0, 2, 3).reshape(S * P, 3, 3)               # record = s*P + p
    return host_math.pack_params(K_inv=K_inv, homs=homs, depths=d.repeat_interleave(P)), P


def warp_params(H_src_tgt, K_inv, G, depth_S):
    """Host image of d_params for mpf_warp_composite."""
    d = host_math._cpu32(depth_S).reshape(-1)
    return host_math.pack_params(K_inv=K_inv, G=G, homs=host_math._cpu32(H_src_tgt).reshape(d.numel(), 3, 3), depths=d)


@_on_device
def src_blend_flow(mpi_S4HW, img_3HW, K_inv=None, depth_S=None, homs_tgt_src=None, flow_clip=200.0,
                   want_rgba=True, want_planar=False, want_tacc=False, out_rgba=None, out_flows=None,
                   dparams=None, P=None, src_u8=None, obj_mask=None, quads=None, quads_complement=None, cum_mask=None):
    """Stage A + C.  homs_tgt_src: None or [P,S,3,3] CPU (P <= 2), or pass a pre-uploaded `dparams` + P.
    Fused by-products (preallocated outputs, optional): src_u8 [H,W,3] u8 BGR source frame; quads / quads_complement
    [H,W,4] = mask_quads(obj_mask, False / True).  cum_mask [S,H,W]: `mpi` is the RAW decoder output and the network's
    activation epilogue (sigmoid / relu(x*cum_mask)+1e-4) is fused into this pass.  Returns dict(rgba, rgb_planar, tacc, flows)."""
    lib = _lib.load()
    mpi = _dev(mpi_S4HW, "mpi")
    S, C, H, W = mpi.shape
    assert C == 4
    img = _dev(img_3HW, "img").reshape(3, H, W)
    if dparams is None:
        params, P = blend_flow_params(K_inv, depth_S, homs_tgt_src)
        dparams = upload_params(params, mpi.device)
    rgba = out_rgba if out_rgba is not None else (torch.empty((S, H, W, 4), dtype=_f32, device=mpi.device) if want_rgba else None)
    planar = torch.empty((S, 3, H, W), dtype=_f32, device=mpi.device) if want_planar else None
    tacc = torch.empty((S, H, W), dtype=_f32, device=mpi.device) if want_tacc else None
    flows = out_flows if out_flows is not None else (torch.empty((P, 2, H, W), dtype=_f32, device=mpi.device) if P else None)
    _lib.check(lib.mpf_src_blend_flow(_ptr(mpi), _ptr(img), _ptr(dparams), P, S, H, W, float(flow_clip), _ptr(rgba),
                                      _ptr(planar), _ptr(tacc), _ptr(flows), _ptr(src_u8),
                                      _ptr(_dev(obj_mask, "obj_mask").reshape(H, W)) if obj_mask is not None else None,
                                      _ptr(quads), _ptr(quads_complement),
                                      _ptr(_dev(cum_mask, "cum_mask")) if cum_mask is not None else None, _stream()), "mpf_src_blend_flow")
    return dict(rgba=rgba, rgb_planar=planar, tacc=tacc, flows=flows)


@_on_device
def alloc_rgba_stack(S, H, W, device):
    """Interleaved [S,H,W,4] stack followed by (W+2) zeroed texels: lets Stage B read the east/south bilinear taps at fixed
    +16 / +row-byte offsets (`interleaved=2`); taps that fall outside the image carry weight exactly 0."""
    n = S * H * W * 4
    store = torch.zeros(n + (W + 2) * 4, dtype=_f32, device=device)
    return store[:n].view(S, H, W, 4)


@_on_device
def mask_quads(obj_mask_HW, complement=False):
    lib = _lib.load()
    m = _dev(obj_mask_HW, "obj_mask")
    H, W = m.shape[-2:]
    m = m.reshape(H, W)
    q = torch.empty((H, W, 4), dtype=_f32, device=m.device)
    _lib.check(lib.mpf_build_mask_quads(_ptr(m), int(bool(complement)), H, W, _ptr(q), _stream()), "mpf_build_mask_quads")
    return q


@_on_device
def warp_composite(rgba, quads, H_src_tgt=None, K_inv=None, G=None, depth_S=None, interleaved=True, want_depth=True,
                   want_tgt_mask=True, dparams=None, out=None):
    """Stage B.  rgba [S,H,W,4] (interleaved) or [S,4,H,W]; quads from mask_quads() or None.  Either pass the small
    matrices or a pre-uploaded `dparams`.  `out`: optional dict of preallocated outputs to reuse.
    Returns dict(rgb [3,H,W], depth [H,W], objmask [H,W] | None, tgt_mask [H,W])."""
    lib = _lib.load()
    a = _dev(rgba, "rgba")
    if interleaved == 2:      # caller guarantees >= (W+1) texels of finite padding after the last plane (alloc_rgba_stack)
        assert a.untyped_storage().nbytes() - a.storage_offset() * 4 >= a.numel() * 4 + (a.shape[2] + 1) * 16
    if interleaved:
        S, H, W, C = a.shape
    else:
        S, C, H, W = a.shape
    assert C == 4
    if dparams is None:
        dparams = upload_params(warp_params(H_src_tgt, K_inv, G, depth_S), a.device)
    q = _dev(quads, "mask quads") if quads is not None else None
    u8 = out.get("rgb_u8") if out is not None else None
    if out is not None:
        rgb, depth, om, tm = out["rgb"], out.get("depth"), out.get("objmask"), out.get("tgt_mask")
    else:
        rgb = torch.empty((3, H, W), dtype=_f32, device=a.device)
        depth = torch.empty((H, W), dtype=_f32, device=a.device) if want_depth else None
        om = torch.empty((H, W), dtype=_f32, device=a.device) if q is not None else None
        tm = torch.empty((H, W), dtype=_f32, device=a.device) if want_tgt_mask else None
    _lib.check(lib.mpf_warp_composite(_ptr(a), int(interleaved), _ptr(q), _ptr(dparams), S, H, W, _ptr(rgb),
                                      _ptr(depth), _ptr(om), _ptr(tm), _ptr(u8), _stream()), "mpf_warp_composite")
    return dict(rgb=rgb, depth=depth, objmask=om, tgt_mask=tm, rgb_u8=u8)


@_on_device
def warp_composite_split(rgb_S3HW, sigma_S1HW, quads, H_src_tgt=None, K_inv=None, G=None, depth_S=None, want_depth=True, want_tgt_mask=True,
                         dparams=None, out=None):
    """Stage B on the two channel-planar tensors render_novel_view_dynamic receives (mpi_all_rgb_src [S,3,H,W], mpi_all_sigma_src
    [S,1,H,W] | [S,H,W]), read in place: no concatenation, no repack (mpf_warp_composite_split).  Returns warp_composite's dict."""
    lib = _lib.load()
    rgb_in = _dev(rgb_S3HW, "rgb stack")
    S, C, H, W = rgb_in.shape
    assert C == 3
    sig = _dev(sigma_S1HW, "sigma stack").reshape(S, H, W)
    if dparams is None:
        dparams = upload_params(warp_params(H_src_tgt, K_inv, G, depth_S), rgb_in.device)
    q = _dev(quads, "mask quads") if quads is not None else None
    u8 = out.get("rgb_u8") if out is not None else None
    if out is not None:
        rgb, depth, om, tm = out["rgb"], out.get("depth"), out.get("objmask"), out.get("tgt_mask")
    else:
        rgb = torch.empty((3, H, W), dtype=_f32, device=rgb_in.device)
        depth = torch.empty((H, W), dtype=_f32, device=rgb_in.device) if want_depth else None
        om = torch.empty((H, W), dtype=_f32, device=rgb_in.device) if q is not None else None
        tm = torch.empty((H, W), dtype=_f32, device=rgb_in.device) if want_tgt_mask else None
    _lib.check(lib.mpf_warp_composite_split(_ptr(rgb_in), _ptr(sig), _ptr(q), _ptr(dparams), S, H, W, _ptr(rgb), _ptr(depth), _ptr(om), _ptr(tm),
                                            _ptr(u8), _stream()), "mpf_warp_composite_split")
    return dict(rgb=rgb, depth=depth, objmask=om, tgt_mask=tm, rgb_u8=u8)


@_on_device
def src_flow(sigma_S1HW, K_inv, depth_S, homs_tgt_src, flow_clip=200.0):
    """Stage C alone (mpf_src_flow): volume-rendered flows [P,2,H,W] of P <= 2 poses from a bare sigma tensor [S,1,H,W] | [S,H,W]."""
    lib = _lib.load()
    sig = _dev(sigma_S1HW, "sigma stack")
    S, H, W = sig.shape[0], sig.shape[-2], sig.shape[-1]
    sig = sig.reshape(S, H, W)
    params, P = blend_flow_params(K_inv, depth_S, homs_tgt_src)
    dparams = upload_params(params, sig.device)
    flows = torch.empty((P, 2, H, W), dtype=_f32, device=sig.device)
    _lib.check(lib.mpf_src_flow(_ptr(sig), _ptr(dparams), P, S, H, W, float(flow_clip), _ptr(flows), _stream()), "mpf_src_flow")
    return flows


@_on_device
def src_flow_hard(sigma_or_stack, K_inv, depth_S, homs_tgt_src, flow_clip=200.0):
    """hard_flow=True in one pass (mpf_src_flow_hard): [P,2,H,W] flows of the arg-max-weight plane.  sigma_or_stack: a bare sigma tensor [S,1,H,W] | [S,H,W],
    or the [S,4,H,W] stack (its sigma planes are read in place)."""
    lib = _lib.load()
    t = _dev(sigma_or_stack, "sigma stack")
    S, H, W = t.shape[0], t.shape[-2], t.shape[-1]
    N = H * W
    if t.dim() == 4 and t.shape[1] == 4:
        ptr, stride = t.data_ptr() + 3 * N * 4, 4 * N
    else:
        t = t.reshape(S, H, W)
        ptr, stride = t.data_ptr(), N
    params, P = blend_flow_params(K_inv, depth_S, homs_tgt_src)
    dparams = upload_params(params, t.device)
    flows = torch.empty((P, 2, H, W), dtype=_f32, device=t.device)
    _lib.check(lib.mpf_src_flow_hard(ctypes.c_void_p(ptr), stride, _ptr(dparams), P, S, H, W, float(flow_clip), _ptr(flows), _stream()), "mpf_src_flow_hard")
    return flows


@_on_device
def warp_composite_views(rgba, views, interleaved=2):
    """Stage B for several views of one interleaved stack in ONE launch (mpf_warp_composite_views): the stack crosses the HBM
    interface once instead of once per view.  views: list of dicts(dparams=, quads= | None, out=dict(rgb, objmask?, depth?,
    tgt_mask?, rgb_u8?)) - every buffer preallocated.  Bit-identical to len(views) warp_composite calls."""
    lib = _lib.load()
    a = _dev(rgba, "rgba")
    S, H, W, C = a.shape
    assert C == 4 and interleaved in (1, 2) and 1 <= len(views) <= _lib.MAX_VIEWS
    if interleaved == 2:
        assert a.untyped_storage().nbytes() - a.storage_offset() * 4 >= a.numel() * 4 + (W + 1) * 16
    arr = (_lib.MpfWarpView * len(views))()
    for i, v in enumerate(views):
        o = v["out"]
        q = v.get("quads")
        for t in [v["dparams"], q] + [o.get(k) for k in ("rgb", "depth", "objmask", "tgt_mask", "rgb_u8")]:
            assert t is None or (t.is_cuda and t.is_contiguous() and t.device == a.device)
        arr[i] = _lib.MpfWarpView(v["dparams"].data_ptr(), q.data_ptr() if q is not None else None, o["rgb"].data_ptr(),
                                  *[(o[k].data_ptr() if o.get(k) is not None else None) for k in ("depth", "objmask", "tgt_mask", "rgb_u8")])
    _lib.check(lib.mpf_warp_composite_views(_ptr(a), int(interleaved), arr, len(views), S, H, W, _stream()), "mpf_warp_composite_views")
    return [v["out"] for v in views]


def _view_array(views, device):
    arr = (_lib.MpfWarpView * len(views))()
    for i, v in enumerate(views):
        o = v["out"]
        q = v.get("quads")
        for t in [v["dparams"], q] + [o.get(k) for k in ("rgb", "depth", "objmask", "tgt_mask", "rgb_u8")]:
            assert t is None or (t.is_cuda and t.is_contiguous() and t.device == device)
        arr[i] = _lib.MpfWarpView(v["dparams"].data_ptr(), q.data_ptr() if q is not None else None, o["rgb"].data_ptr(),
                                  *[(o[k].data_ptr() if o.get(k) is not None else None) for k in ("depth", "objmask", "tgt_mask", "rgb_u8")])
    return arr


@_on_device
def warp_views_and_blend_next(rgba, views, mpi_next, img_next, dparams_next, P, out_rgba_next, out_flows_next=None, flow_clip=200.0,
                              src_u8_next=None, obj_mask_next=None, quads_next=None, quads_complement_next=None, cum_mask_next=None,
                              merge_prev=None):
    """Stage B of one image (all `views` of the tail-padded stack `rgba`, as warp_composite_views) and Stage A+C of the NEXT image
    (as src_blend_flow with preallocated outputs) in ONE launch whose grid interleaves the two kinds of workgroups
    (mpf_warp_views_and_blend_next).  Bit-identical to the two separate calls; every *_next buffer must be distinct from what the
    views read or write.  merge_prev: merge_args(...) of an EARLIER pair, merged by the Stage A+C role as a per-pixel prologue
    (mpf_warp_views_blend_next_merge_prev); its flows may be `out_flows_next` itself."""
    lib = _lib.load()
    a = _dev(rgba, "rgba")
    S, H, W, C = a.shape
    assert C == 4 and 1 <= len(views) <= _lib.MAX_VIEWS
    assert a.untyped_storage().nbytes() - a.storage_offset() * 4 >= a.numel() * 4 + (W + 1) * 16
    mpi = _dev(mpi_next, "mpi_next")
    assert tuple(mpi.shape) == (S, 4, H, W) and out_rgba_next.is_contiguous() and tuple(out_rgba_next.shape) == (S, H, W, 4)
    assert out_rgba_next.data_ptr() != a.data_ptr()
    img = _dev(img_next, "img_next").reshape(3, H, W)
    arr = _view_array(views, a.device)
    om = _dev(obj_mask_next, "obj_mask_next").reshape(H, W) if obj_mask_next is not None else None
    cm = _dev(cum_mask_next, "cum_mask_next") if cum_mask_next is not None else None
    _lib.check(lib.mpf_warp_views_blend_next_merge_prev(_ptr(a), arr, len(views), _ptr(mpi), _ptr(img), _ptr(dparams_next), int(P), float(flow_clip),
                                                        _ptr(out_rgba_next), _ptr(out_flows_next), _ptr(src_u8_next), _ptr(om), _ptr(quads_next),
                                                        _ptr(quads_complement_next), _ptr(cm), S, H, W,
                                                        ctypes.byref(merge_prev) if merge_prev is not None else None, _stream()),
               "mpf_warp_views_blend_next_merge_prev")
    return [v["out"] for v in views]


def merge_args(frame, frame_dyn, mask, mask_dyn, flow, flow_dyn, obj_mask, thresh, out, obj_mask_stride=1):
    """mpf_merge's arguments as the struct a pair launch takes (warp_views_and_blend_next(merge_prev=...)) or mpf_merge_ex.  All tensors fp32
    contiguous on the device, out = (flow_mix [H,W,2] f32, frame_mix [H,W,3] u8, fill_mask [H,W] u8) on the same device; the caller keeps them
    alive until the launch was issued.  obj_mask_stride = 4: `obj_mask` is a mask-quad buffer [H,W,4] whose .x is the object mask."""
    import numpy as np
    for t in (frame, frame_dyn, mask, mask_dyn, flow, flow_dyn, obj_mask):
        assert t.is_cuda and t.dtype == _f32 and t.is_contiguous()
    _, H, W = frame.shape
    assert obj_mask_stride in (1, 4) and obj_mask.numel() == H * W * obj_mask_stride, (obj_mask_stride, tuple(obj_mask.shape))
    fm, fr, fi = out
    assert fm.is_cuda and fm.dtype == _f32 and fm.is_contiguous() and tuple(fm.shape) == (H, W, 2), "flow_mix must be f32 [H,W,2], contiguous"
    assert fr.is_cuda and fr.dtype == torch.uint8 and fr.is_contiguous() and tuple(fr.shape) == (H, W, 3), "frame_mix must be u8 [H,W,3], contiguous"
    assert fi.is_cuda and fi.dtype == torch.uint8 and fi.is_contiguous() and tuple(fi.shape) == (H, W), "fill_mask must be u8 [H,W], contiguous"
    assert fm.device == fr.device == fi.device == frame.device, "merge outputs must live on the device of the views"
    return _lib.MpfMergeArgs(frame.data_ptr(), frame_dyn.data_ptr(), mask.data_ptr(), mask_dyn.data_ptr(), flow.data_ptr(), flow_dyn.data_ptr(),
                             obj_mask.data_ptr(), float(np.float32(thresh)), fm.data_ptr(), fr.data_ptr(), fi.data_ptr(), int(obj_mask_stride))


def pair_slab(H, W, device):
    """One contiguous device buffer holding a pair's three products - flow_mix [H,W,2] f32 | frame_mix [H,W,3] u8 | fill_mask [H,W] u8 (12 H W bytes) -
    so that they leave the GPU in ONE device-to-host copy (io_formats.OutputRing.submit_pair_fill(slab=...)).  -> (slab u8 [12 H W], (flow_mix, frame_mix, fill_mask) views)"""
    n = H * W
    slab = torch.empty(12 * n, dtype=torch.uint8, device=device)
    return slab, slab_views(slab, H, W)


def slab_views(slab, H, W):
    n = H * W
    return (slab[:8 * n].view(torch.float32).view(H, W, 2), slab[8 * n:11 * n].view(H, W, 3), slab[11 * n:12 * n].view(H, W))


@_on_device
def merge(frame, frame_dyn, mask, mask_dyn, flow, flow_dyn, obj_mask, thresh=0.99, out=None, obj_mask_stride=1):
    """Stage D.  out: optional preallocated (flow_mix [H,W,2] f32, frame_mix [H,W,3] u8, fill_mask [H,W] u8); default: views of one pair_slab.
    obj_mask_stride = 4: `obj_mask` is a mask-quad buffer [H,W,4] whose .x is the object mask (mpf_merge_ex)."""
    lib = _lib.load()
    frame = _dev(frame, "frame")
    _, H, W = frame.shape
    dev = frame.device
    if out is not None:
        flow_mix, frame_mix, fill = out
    else:
        _, (flow_mix, frame_mix, fill) = pair_slab(H, W, dev)
    if obj_mask_stride != 1:
        import ctypes
        a = merge_args(frame, _dev(frame_dyn, "frame_dyn").reshape(3, H, W), _dev(mask, "mask").reshape(H, W), _dev(mask_dyn, "mask_dyn").reshape(H, W),
                       _dev(flow, "flow").reshape(2, H, W), _dev(flow_dyn, "flow_dyn").reshape(2, H, W), _dev(obj_mask, "obj_mask"), thresh,
                       (flow_mix, frame_mix, fill), obj_mask_stride=obj_mask_stride)
        _lib.check(lib.mpf_merge_ex(ctypes.byref(a), H, W, _stream()), "mpf_merge_ex")
        return flow_mix, frame_mix, fill
    args = [_dev(frame_dyn, "frame_dyn").reshape(3, H, W), _dev(mask, "mask").reshape(H, W),
            _dev(mask_dyn, "mask_dyn").reshape(H, W), _dev(flow, "flow").reshape(2, H, W),
            _dev(flow_dyn, "flow_dyn").reshape(2, H, W), _dev(obj_mask, "obj_mask").reshape(H, W)]
    import numpy as np
    _lib.check(lib.mpf_merge(_ptr(frame), *[_ptr(a) for a in args], float(np.float32(thresh)), H, W, _ptr(flow_mix),
                             _ptr(frame_mix), _ptr(fill), _stream()), "mpf_merge")
    return flow_mix, frame_mix, fill


@_on_device
def merge_depth_ordered(frame, frame_dyn, mask, mask_dyn, depth, depth_dyn, thresh=0.99, out=None, want_depth_mask=False):
    """The depth-ordered frame of the reference's older module ("utils/utils copy.py":278-303): Stage D's frame_mix, except that where both
    layers cover the pixel (both masks non-zero) and depth > depth_dyn the dynamic layer's pixel is taken.
    -> frame_mix_depth [H,W,3] u8 BGR (, depth_mask [H,W] u8 when want_depth_mask)."""
    import numpy as np
    lib = _lib.load()
    frame = _dev(frame, "frame")
    _, H, W = frame.shape
    dev = frame.device
    fmd = out if out is not None else torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    dm = torch.empty((H, W), dtype=torch.uint8, device=dev) if want_depth_mask else None
    args = [_dev(frame_dyn, "frame_dyn").reshape(3, H, W), _dev(mask, "mask").reshape(H, W), _dev(mask_dyn, "mask_dyn").reshape(H, W),
            _dev(depth, "depth").reshape(H, W), _dev(depth_dyn, "depth_dyn").reshape(H, W)]
    _lib.check(lib.mpf_merge_depth_ordered(_ptr(frame), *[_ptr(a) for a in args], float(np.float32(thresh)), H, W, _ptr(fmd), _ptr(dm),
                                           _stream()), "mpf_merge_depth_ordered")
    return (fmd, dm) if want_depth_mask else fmd


@_on_device
def fill_holes(img_HW3_u8, hole_HW_u8, out=None, hole_out=None, workspace=None):
    """Built-in deterministic hole fill (onion peel; NOT OpenCV's algorithm - see DESIGN.md, row A13).  Stream-ordered, no
    host synchronisation.  Returns the filled copy (`out`); `hole_out` (optional) receives the holes still open."""
    lib = _lib.load()
    src = _dev(img_HW3_u8, "img", torch.uint8)
    H, W, _ = src.shape
    out = torch.empty_like(src) if out is None else out
    out.copy_(src)
    hole = torch.empty((H, W), dtype=torch.uint8, device=src.device) if hole_out is None else hole_out
    hole.copy_(_dev(hole_HW_u8, "hole", torch.uint8))
    need = int(lib.mpf_fill_holes_workspace(H, W))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=src.device)
    _lib.check(lib.mpf_fill_holes(_ptr(out), _ptr(hole), H, W, _ptr(workspace), need, _stream()), "mpf_fill_holes")
    return out


@_on_device
def prepare_inputs(rgb_u8=None, disp_u8=None, ids_u8=None, obj_index=0, size=None, out=None):
    """Input stage in one launch (mpf_prepare_inputs): uploaded u8 buffers -> resized float tensors, bit-identical to the
    reference's ToTensor / `/255` / (ids == k).float() followed by F.interpolate(bilinear, align_corners=True).
    rgb_u8 [h,w,3], disp_u8 [h,w], ids_u8 [h,w] on the device (any subset); size = (H, W).
    Returns dict(image [3,H,W], disp [H,W], mask [H,W]) (None for absent inputs); `out` may carry preallocated tensors."""
    lib = _lib.load()
    H, W = size
    first = next(t for t in (rgb_u8, disp_u8, ids_u8) if t is not None)
    h, w = first.shape[:2]
    dev = first.device
    out = dict(out or {})
    rgb = _dev(rgb_u8, "rgb", torch.uint8) if rgb_u8 is not None else None
    dsp = _dev(disp_u8, "disp", torch.uint8) if disp_u8 is not None else None
    ids = _dev(ids_u8, "ids", torch.uint8) if ids_u8 is not None else None
    for t in (rgb, dsp, ids):
        assert t is None or tuple(t.shape[:2]) == (h, w)
    image = (out.get("image") if out.get("image") is not None else torch.empty((3, H, W), dtype=_f32, device=dev)) if rgb is not None else None
    disp = (out.get("disp") if out.get("disp") is not None else torch.empty((H, W), dtype=_f32, device=dev)) if dsp is not None else None
    mask = (out.get("mask") if out.get("mask") is not None else torch.empty((H, W), dtype=_f32, device=dev)) if ids is not None else None
    _lib.check(lib.mpf_prepare_inputs(_ptr(rgb), _ptr(dsp), _ptr(ids), int(obj_index), h, w, H, W, _ptr(image), _ptr(disp), _ptr(mask), _stream()),
               "mpf_prepare_inputs")
    return dict(image=image, disp=disp, mask=mask)


INPAINT_NS, INPAINT_TELEA = 0, 1          # cv2.INPAINT_NS / cv2.INPAINT_TELEA (MPF_INPAINT_*)


def inpaint_host(img_u8, mask_u8, radius=3, method=INPAINT_NS, out=None):
    """The reference's hole filling, cv2.inpaint(img, mask, radius, method) (utils/utils.py:284-286 NS, moving_obj.py:162 TELEA),
    as restated in libmpiflow_hip.so (mpf_inpaint_host): HOST numpy arrays in and out, like the reference's call; synchronous;
    ctypes releases the GIL, so frames fill in parallel on the generator's writer threads.  img u8 [H,W,3] | [H,W], mask u8 [H,W]."""
    import numpy as np
    lib = _lib.load()
    img = np.ascontiguousarray(img_u8, dtype=np.uint8)
    H, W = img.shape[:2]
    C = 1 if img.ndim == 2 else img.shape[2]
    mask = np.ascontiguousarray(mask_u8, dtype=np.uint8).reshape(H, W)
    if out is None:
        out = np.empty_like(img)
    assert out.shape == img.shape and out.dtype == np.uint8 and out.flags.c_contiguous and out is not img
    _lib.check(lib.mpf_inpaint_host(img.ctypes.data, mask.ctypes.data, H, W, C, float(radius), int(method), out.ctypes.data), "mpf_inpaint_host")
    return out


PAIR_STATS_SLICES = 64          # MPF_PAIR_STATS_SLICES


@_on_device
def pair_stats(flow_mix_HW2, fill_mask_HW, out4):
    """mpf_pair_stats: per-slice {sum |flow|, hole px, max |flow|, max(-flow)} of one pair into out4 ([64,4] float64 on the device),
    stream-ordered"""
    lib = _lib.load()
    flow = _dev(flow_mix_HW2, "flow_mix")
    fill = _dev(fill_mask_HW, "fill_mask", torch.uint8)
    H, W = fill.shape
    assert out4.dtype == torch.float64 and out4.numel() == 4 * PAIR_STATS_SLICES and out4.is_contiguous()
    _lib.check(lib.mpf_pair_stats(_ptr(flow), _ptr(fill), H, W, _ptr(out4), _stream()), "mpf_pair_stats")
    return out4


@_on_device
def png_scanlines(img_HW3_bgr_u8, out=None):
    """[H,W,3] u8 BGR on the device -> PNG scanlines u8 [H, 1+3W] (filter "Up", RGB order) for io_formats.png_from_scanlines"""
    lib = _lib.load()
    img = _dev(img_HW3_bgr_u8, "img", torch.uint8)
    H, W, _ = img.shape
    out = torch.empty((H, 3 * W + 1), dtype=torch.uint8, device=img.device) if out is None else out
    _lib.check(lib.mpf_png_filter_up(_ptr(img), H, W, _ptr(out), _stream()), "mpf_png_filter_up")
    return out


@_on_device
def to_u8_bgr(img_3HW):
    lib = _lib.load()
    img = _dev(img_3HW, "img")
    _, H, W = img.shape
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=img.device)
    _lib.check(lib.mpf_to_u8_bgr(_ptr(img), H, W, _ptr(out), _stream()), "mpf_to_u8_bgr")
    return out


# ---- generic ops ----------------------------------------------------------------------------------------------------

@_on_device
def src_xyz(K_inv, depth_S, H, W, device):
    lib = _lib.load()
    d = host_math._cpu32(depth_S).reshape(-1)
    S = d.numel()
    dparams = upload_params(host_math.pack_params(K_inv=K_inv, depths=d), device)
    out = torch.empty((S, 3, H, W), dtype=_f32, device=device)
    _lib.check(lib.mpf_src_xyz(_ptr(dparams), S, H, W, _ptr(out), _stream()), "mpf_src_xyz")
    return out


@_on_device
def transform_xyz(G, xyz_S3N):
    lib = _lib.load()
    xyz = _dev(xyz_S3N, "xyz")
    S = xyz.shape[0]
    N = xyz[0, 0].numel()
    dparams = upload_params(host_math.pack_params(G=G, records=1), xyz.device)
    out = torch.empty_like(xyz)
    _lib.check(lib.mpf_transform_xyz(_ptr(dparams), _ptr(xyz), S, N, _ptr(out), _stream()), "mpf_transform_xyz")
    return out


@_on_device
def homography_sample(src_SCHW, H_src_tgt, want_flow=True):
    lib = _lib.load()
    src = _dev(src_SCHW, "src")
    S, C, H, W = src.shape
    dparams = upload_params(host_math.pack_params(homs=host_math._cpu32(H_src_tgt).reshape(S, 3, 3)), src.device)
    tgt = torch.empty_like(src)
    valid = torch.empty((S, H, W), dtype=torch.uint8, device=src.device)
    flow = torch.empty((S, H, W, 2), dtype=_f32, device=src.device) if want_flow else None
    _lib.check(lib.mpf_homography_sample(_ptr(src), _ptr(dparams), S, C, H, W, _ptr(tgt), _ptr(valid), _ptr(flow), _stream()),
               "mpf_homography_sample")
    return tgt, valid.to(torch.bool), flow


@_on_device
def homography_flow(H_tgt_src, H, W, device):
    lib = _lib.load()
    hts = host_math._cpu32(H_tgt_src).reshape(-1, 3, 3)
    S = hts.shape[0]
    dparams = upload_params(host_math.pack_params(homs=hts), device)
    flow = torch.empty((S, H, W, 2), dtype=_f32, device=device)
    _lib.check(lib.mpf_homography_flow(_ptr(dparams), S, H, W, _ptr(flow), _stream()), "mpf_homography_flow")
    return flow


@_on_device
def volume_render(rgb_S3N, sigma_SN, xyz_S3N, extra_SEN=None, hard=False, want_tacc=True, want_weights=True):
    """Generic plane_volume_rendering on materialised tensors.  Trailing dims are flattened to N."""
    lib = _lib.load()
    xyz = _dev(xyz_S3N, "xyz")
    S = xyz.shape[0]
    tail = tuple(xyz.shape[2:])
    N = xyz[0, 0].numel()
    sigma = _dev(sigma_SN, "sigma")
    rgb = _dev(rgb_S3N, "rgb") if rgb_S3N is not None else None
    extra = _dev(extra_SEN, "extra") if extra_SEN is not None else None
    E = 0 if extra is None else extra.shape[1]
    dev = xyz.device
    out = dict(rgb=torch.empty((3,) + tail, dtype=_f32, device=dev) if rgb is not None else None,
               depth=torch.empty(tail, dtype=_f32, device=dev),
               tacc=torch.empty((S,) + tail, dtype=_f32, device=dev) if want_tacc else None,
               weights=torch.empty((S,) + tail, dtype=_f32, device=dev) if want_weights else None,
               extra=torch.empty((E,) + tail, dtype=_f32, device=dev) if E else None)
    _lib.check(lib.mpf_volume_render(_ptr(rgb), _ptr(sigma), _ptr(xyz), S, N, _ptr(out["rgb"]), _ptr(out["depth"]),
                                     _ptr(out["tacc"]), _ptr(out["weights"]), _ptr(extra), E, _ptr(out["extra"]),
                                     int(bool(hard)), _stream()), "mpf_volume_render")
    return out


@_on_device
def weighted_sum(weights_SN, values_SCN=None):
    """cascade-sum over S of weights (* values).  weights [S,*tail], values [S,C,*tail] -> [C,*tail] ([1,*tail] if None)"""
    lib = _lib.load()
    w = _dev(weights_SN, "weights")
    S = w.shape[0]
    tail = tuple(w.shape[1:])
    N = w[0].numel()
    v = _dev(values_SCN, "values") if values_SCN is not None else None
    C = v.shape[1] if v is not None else 1
    out = torch.empty((C,) + tail, dtype=_f32, device=w.device)
    _lib.check(lib.mpf_weighted_sum(_ptr(w), _ptr(v), S, C, N, _ptr(out), _stream()), "mpf_weighted_sum")
    return out


# ---- depth -> flow, forward warp -----------------------------------------------------------------------------------

@_on_device
def alpha_composite(alpha_SN, values_SCN=None, want_weights=True, want_cumprod_eps=False):
    """alpha_composition (mpi_rendering.py:42-59) -> dict(out [C,N] | None, weights [S,N] | None, cumprod_eps [S,N] | None)"""
    lib = _lib.load()
    al = _dev(alpha_SN, "alpha")
    S, N = al.shape[0], al[0].numel()
    al = al.reshape(S, N)
    dev = al.device
    vals = out = None
    C = 1
    if values_SCN is not None:
        vals = _dev(values_SCN, "values")
        C = vals.shape[1]
        vals = vals.reshape(S, C, N)
        out = torch.empty((C, N), dtype=_f32, device=dev)
    w = torch.empty((S, N), dtype=_f32, device=dev) if want_weights else None
    ce = torch.empty((S, N), dtype=_f32, device=dev) if want_cumprod_eps else None
    _lib.check(lib.mpf_alpha_composite(_ptr(al), _ptr(vals), S, C, ctypes.c_int64(N), _ptr(out), _ptr(w), _ptr(ce), _stream()), "mpf_alpha_composite")
    return dict(out=out, weights=w, cumprod_eps=ce)


@_on_device
def disp_to_depth(disp):
    lib = _lib.load()
    d = _dev(disp, "disp")
    out = torch.empty_like(d)
    _lib.check(lib.mpf_disp_to_depth(_ptr(d), d.numel(), _ptr(out), _stream()), "mpf_disp_to_depth")
    return out


@_on_device
def backproject_project(depth_HW, inv_K33, P34):
    lib = _lib.load()
    depth = _dev(depth_HW, "depth")
    H, W = depth.shape[-2:]
    depth = depth.reshape(H, W)
    ik = host_math._cpu32(inv_K33).reshape(9).contiguous()
    P = host_math._cpu32(P34).reshape(12).contiguous()
    pix = torch.empty((H, W, 2), dtype=_f32, device=depth.device)
    z = torch.empty((H, W), dtype=_f32, device=depth.device)
    _lib.check(lib.mpf_backproject_project(_ptr(depth), ctypes.c_void_p(ik.data_ptr()), ctypes.c_void_p(P.data_ptr()), H, W,
                                           _ptr(pix), _ptr(z), _stream()), "mpf_backproject_project")
    return pix, z


@_on_device
def backproject(depth_HW, inv_K33):
    """BackprojectDepth.forward -> [4, H*W] camera points (rows X, Y, Z, 1)"""
    lib = _lib.load()
    depth = _dev(depth_HW, "depth")
    H, W = depth.shape[-2:]
    ik = host_math._cpu32(inv_K33).reshape(9).contiguous()
    cam = torch.empty((4, H * W), dtype=_f32, device=depth.device)
    _lib.check(lib.mpf_backproject(_ptr(depth.reshape(H, W)), ctypes.c_void_p(ik.data_ptr()), H, W, _ptr(cam), _stream()), "mpf_backproject")
    return cam


@_on_device
def project3d(points_4N, P34, H, W, eps=1e-7):
    """Project3D.forward on homogeneous points [4, H*W] -> (pix [H,W,2] normalised, z [H*W])"""
    lib = _lib.load()
    pts = _dev(points_4N, "points").reshape(4, H * W)
    P = host_math._cpu32(P34).reshape(12).contiguous()
    pix = torch.empty((H, W, 2), dtype=_f32, device=pts.device)
    z = torch.empty((H * W,), dtype=_f32, device=pts.device)
    _lib.check(lib.mpf_project3d(_ptr(pts), ctypes.c_void_p(P.data_ptr()), float(eps), H, W, _ptr(pix), _ptr(z), _stream()), "mpf_project3d")
    return pix, z


@_on_device
def select_truncate(p_static, z_static, p_obj, z_obj, inst_HW):
    lib = _lib.load()
    inst = _dev(inst_HW, "instance mask")
    H, W = inst.shape[-2:]
    dev = inst.device
    p1 = torch.empty((H, W, 2), dtype=_f32, device=dev)
    z1 = torch.empty((H, W), dtype=_f32, device=dev)
    sx = torch.empty((H, W), dtype=torch.int64, device=dev)
    sy = torch.empty((H, W), dtype=torch.int64, device=dev)
    fl = torch.empty((H, W, 2), dtype=_f32, device=dev)
    _lib.check(lib.mpf_select_truncate(_ptr(_dev(p_static, "p_static")), _ptr(_dev(z_static, "z_static")),
                                       _ptr(_dev(p_obj, "p_obj")), _ptr(_dev(z_obj, "z_obj")), _ptr(inst.reshape(H, W)), H, W,
                                       _ptr(p1), _ptr(z1), _ptr(sx), _ptr(sy), _ptr(fl), _stream()), "mpf_select_truncate")
    return p1, z1, sx, sy, fl


@_on_device
def moving_object_project(disp_HW, inv_K33, P_static34, P_obj34, inst_HW):
    """Fused moving_obj.py:29-124: -> (p1 [H,W,2], z1 [H,W], safe_x, safe_y int64 [H,W], flow01 [H,W,2])"""
    lib = _lib.load()
    disp = _dev(disp_HW, "disp")
    H, W = disp.shape[-2:]
    dev = disp.device
    inst = _dev(inst_HW, "instance mask").reshape(H, W)
    ik = host_math._cpu32(inv_K33).reshape(9).contiguous()
    Ps = host_math._cpu32(P_static34).reshape(12).contiguous()
    Po = host_math._cpu32(P_obj34).reshape(12).contiguous()
    p1 = torch.empty((H, W, 2), dtype=_f32, device=dev)
    z1 = torch.empty((H, W), dtype=_f32, device=dev)
    sx = torch.empty((H, W), dtype=torch.int64, device=dev)
    sy = torch.empty((H, W), dtype=torch.int64, device=dev)
    fl = torch.empty((H, W, 2), dtype=_f32, device=dev)
    _lib.check(lib.mpf_moving_object_project(_ptr(disp.reshape(H, W)), ctypes.c_void_p(ik.data_ptr()), ctypes.c_void_p(Ps.data_ptr()),
                                             ctypes.c_void_p(Po.data_ptr()), _ptr(inst), H, W, _ptr(p1), _ptr(z1), _ptr(sx), _ptr(sy),
                                             _ptr(fl), _stream()), "mpf_moving_object_project")
    return p1, z1, sx, sy, fl


@_on_device
def forward_warp(src_u8, idx_i64, idy_i64, z_f32, h, w):
    """Device-resident forward splat, byte-identical to the reference's serial C.  -> warped u8 [h,w,5]"""
    lib = _lib.load()
    src = _dev(src_u8, "src", torch.uint8).reshape(-1)
    idx = _dev(idx_i64, "idx", torch.int64).reshape(-1)
    idy = _dev(idy_i64, "idy", torch.int64).reshape(-1)
    z = _dev(z_f32, "z").reshape(-1)
    assert src.numel() == h * w * 3 and idx.numel() == h * w and idy.numel() == h * w and z.numel() == h * w
    ws_bytes = lib.mpf_forward_warp_workspace(h, w)
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=src.device)
    off = (-ws.data_ptr()) % 256
    warped = torch.empty((h, w, 5), dtype=torch.uint8, device=src.device)
    _lib.check(lib.mpf_forward_warp(_ptr(src), _ptr(idx), _ptr(idy), _ptr(z), _ptr(warped), h, w,
                                    ctypes.c_void_p(ws.data_ptr() + off), ws_bytes, _stream()), "mpf_forward_warp")
    return warped


@_on_device
def warp_masks(warped_HW5):
    lib = _lib.load()
    w5 = _dev(warped_HW5, "warped", torch.uint8)
    H, W, _ = w5.shape
    outs = [torch.empty((H, W), dtype=torch.uint8, device=w5.device) for _ in range(5)]
    _lib.check(lib.mpf_warp_masks(_ptr(w5), H, W, *[_ptr(o) for o in outs], _stream()), "mpf_warp_masks")
    return dict(zip(["H", "M", "M'", "P", "H'"], outs))


class MovingObjectBuffers:
    """Preallocated outputs + sort workspace of mpf_moving_object_chain for one (H, W) on one device (28 N bytes of results)."""

    def __init__(self, H, W, device):
        lib = _lib.load()
        dev = torch.device(device)
        u8 = torch.uint8
        self.H, self.W, self.device = H, W, dev
        self.p1 = torch.empty((H, W, 2), dtype=_f32, device=dev)
        self.z1 = torch.empty((H, W), dtype=_f32, device=dev)
        self.safe_x = torch.empty((H, W), dtype=torch.int64, device=dev)
        self.safe_y = torch.empty((H, W), dtype=torch.int64, device=dev)
        self.flow_01 = torch.empty((H, W, 2), dtype=_f32, device=dev)
        self.warped = torch.empty((H, W, 5), dtype=u8, device=dev)
        self.masks = {k: torch.empty((H, W), dtype=u8, device=dev) for k in ("H", "M", "M'", "P", "H'")}
        self.ws_bytes = lib.mpf_forward_warp_workspace(H, W)
        self._ws = torch.empty(self.ws_bytes + 256, dtype=u8, device=dev)
        self.ws_ptr = self._ws.data_ptr() + (-self._ws.data_ptr()) % 256
        m = self.masks
        self.ready = None        # set by pipeline.OverlappedPairRenderer (unordered chain): torch event recorded behind the launches that fill this set
        self.consumed = None     # optional, set by the consumer: torch event the side stream waits for before it rewrites this set
        self.c_out = _lib.MpfMovingObjectOut(self.p1.data_ptr(), self.z1.data_ptr(), self.safe_x.data_ptr(), self.safe_y.data_ptr(),
                                             self.flow_01.data_ptr(), self.warped.data_ptr(), m["H"].data_ptr(), m["M"].data_ptr(),
                                             m["M'"].data_ptr(), m["P"].data_ptr(), m["H'"].data_ptr())

    def as_dict(self):
        return dict(p1=self.p1, z1=self.z1, safe_x=self.safe_x, safe_y=self.safe_y, flow_01=self.flow_01, warped=self.warped, masks=self.masks)


@_on_device
def moving_object_chain(disp_HW, inv_K33, P_static34, P_obj34, inst_HW, src, bufs=None):
    """moving_obj.py:29-150 in one call (mpf_moving_object_chain): projection fused into the forward warp's first sort pass, splat, masks.
    src: the frame that is splatted - uint8 [H,W,3], or float32 [3,H,W] in 0..1 (its uint8 BGR form is splatted, converted on the fly).
    -> MovingObjectBuffers (bufs or a new one)"""
    lib = _lib.load()
    disp = _dev(disp_HW, "disp")
    H, W = disp.shape[-2:]
    inst = _dev(inst_HW, "instance mask")
    as_float = src.dtype != torch.uint8
    src = _dev(src, "src", _f32 if as_float else torch.uint8)
    assert inst.numel() == H * W and src.numel() == H * W * 3 and (not as_float or tuple(src.shape[-3:]) == (3, H, W))
    if bufs is None:
        bufs = MovingObjectBuffers(H, W, disp.device)
    assert (bufs.H, bufs.W) == (H, W) and bufs.device == disp.device
    ik = host_math._cpu32(inv_K33).reshape(9).contiguous()
    Ps = host_math._cpu32(P_static34).reshape(12).contiguous()
    Po = host_math._cpu32(P_obj34).reshape(12).contiguous()
    _lib.check(lib.mpf_moving_object_chain(_ptr(disp), ctypes.c_void_p(ik.data_ptr()), ctypes.c_void_p(Ps.data_ptr()), ctypes.c_void_p(Po.data_ptr()),
                                           _ptr(inst), None if as_float else _ptr(src), _ptr(src) if as_float else None, H, W, ctypes.byref(bufs.c_out), ctypes.c_void_p(bufs.ws_ptr), bufs.ws_bytes,
                                           _stream()), "mpf_moving_object_chain")
    return bufs
