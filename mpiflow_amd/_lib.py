"""ctypes binding of libmpiflow_hip.so (C ABI: include/mpiflow_hip.h).

The shared library is built in-tree by `python __graft_entry__.py` / `make -C mpiflow_amd/csrc` with
`hipcc --offload-arch=gfx950`.  There is NO fallback: if the library is missing or a call fails, this module raises.
"""
import ctypes
import os

# PyTorch bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  It must be the first HIP runtime mapped into the
# process so that libmpiflow_hip.so binds to the one that owns torch's device context and streams; loading ours first
# leaves two half-initialised runtimes ("no ROCm-capable device is detected").
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPIFLOW_HIP_LIB: development hook for same-box A/B timing of two builds of the library (tools/); symbols an older build lacks are skipped
LIB_PATH = os.environ.get("MPIFLOW_HIP_LIB") or os.path.join(_HERE, "libmpiflow_hip.so")

# the WITNESS build of the same sources (-DMPF_WITNESS): + retired kernel variants and timing ablations, selectable through mpf_tune keys the product library
# refuses.  Test / tool infrastructure: the product never loads it on its own (witness() / select_witness() below).
WITNESS_PATH = os.path.join(_HERE, "libmpiflow_hip_witness.so")

_lib = None
_witness = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_i64 = ctypes.c_int64
c_sz = ctypes.c_size_t



class MpfConvArgs(ctypes.Structure):
    """struct MpfConvArgs of include/mpiflow_hip.h (field order and types must match)."""
    _fields_ = [("srcA", c_p), ("srcB", c_p), ("cm", c_p), ("fm", c_p), ("plane_vals", c_p), ("wpack", c_p), ("ep", c_p),
                ("out", c_p),
                ("S", c_i), ("Hin", c_i), ("Win", c_i), ("Hout", c_i), ("Wout", c_i),
                ("CA", c_i), ("CB", c_i), ("HA", c_i), ("WA", c_i),
                ("ct", c_i), ("nchunk", c_i), ("nblk", c_i), ("ncg", c_i), ("Cst", c_i),
                ("loader", c_i), ("epi", c_i), ("stride", c_i), ("pad_mode", c_i),
                ("fparams", c_f * 4), ("wlds", c_i), ("plane_major", c_i), ("bprime_table", c_i), ("pw", c_i)]


class MpfConv2dArgs(ctypes.Structure):
    """struct MpfConv2dArgs of include/mpiflow_hip.h: one fp32 convolution of the single-image encoder / bottleneck."""
    _fields_ = [("src", c_p), ("wpack", c_p), ("scale", c_p), ("shift", c_p), ("residual", c_p), ("out", c_p), ("out_f16", c_p),
                ("Hin", c_i), ("Win", c_i), ("Cin", c_i), ("Hout", c_i), ("Wout", c_i), ("Cout", c_i),
                ("ksize", c_i), ("stride", c_i), ("pad", c_i), ("up", c_i), ("act", c_i), ("slope", c_f)]


class MpfPConvArgs(ctypes.Structure):
    """struct MpfPConvArgs of include/mpiflow_hip.h: one convolution of the parity-grade (fp32 / fp64) producer engine."""
    _fields_ = [("srcA", c_p), ("srcB", c_p), ("wpack", c_p), ("scale", c_p), ("shift", c_p), ("bias", c_p), ("residual", c_p), ("out", c_p),
                ("dtype", c_i), ("S", c_i), ("Hin", c_i), ("Win", c_i), ("Hout", c_i), ("Wout", c_i),
                ("HA", c_i), ("WA", c_i), ("CA", c_i), ("CB", c_i), ("up", c_i), ("shareA", c_i), ("shareB", c_i),
                ("ksize", c_i), ("stride", c_i), ("pad", c_i), ("pad_mode", c_i), ("nblk", c_i), ("Cst", c_i), ("epi", c_i), ("act", c_i), ("slope", ctypes.c_double)]


class MpfWarpView(ctypes.Structure):
    """struct MpfWarpView of include/mpiflow_hip.h: one view of mpf_warp_composite_views (device pointers)."""
    _fields_ = [("d_params", c_p), ("d_mask_quads", c_p), ("d_rgb", c_p), ("d_depth", c_p), ("d_objmask", c_p),
                ("d_tgt_mask", c_p), ("d_rgb_u8_bgr", c_p)]


class MpfMovingObjectOut(ctypes.Structure):
    """struct MpfMovingObjectOut of include/mpiflow_hip.h: the output buffers of mpf_moving_object_chain (device pointers)."""
    _fields_ = [("d_p1", c_p), ("d_z1", c_p), ("d_safe_x", c_p), ("d_safe_y", c_p), ("d_flow01", c_p), ("d_warped", c_p),
                ("d_Hm", c_p), ("d_M", c_p), ("d_Md", c_p), ("d_P", c_p), ("d_Hp", c_p)]


class MpfMergeArgs(ctypes.Structure):
    """struct MpfMergeArgs of include/mpiflow_hip.h: mpf_merge's arguments, for the merge folded into a pair launch."""
    _fields_ = [("d_frame", c_p), ("d_frame_dyn", c_p), ("d_mask", c_p), ("d_mask_dyn", c_p), ("d_flow", c_p), ("d_flow_dyn", c_p),
                ("d_obj_mask", c_p), ("thresh", c_f), ("d_flow_mix", c_p), ("d_frame_mix", c_p), ("d_fill_mask", c_p), ("obj_mask_stride", c_i)]


MAX_VIEWS = 16          # MPF_MAX_VIEWS

# name -> (restype, argtypes); must list every symbol include/mpiflow_hip.h declares (tests/test_capi.py checks)
SIGNATURES = {
    "mpf_version": (c_i, []),
    "mpf_last_error": (ctypes.c_char_p, []),
    "mpf_device_info": (c_i, [c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_sz), ctypes.c_char_p, c_sz]),
    "mpf_stream_create_cu_subset": (c_i, [c_i, c_i, ctypes.POINTER(c_p)]),
    "mpf_stream_destroy": (c_i, [c_p]),
    "mpf_src_blend_flow": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mpf_build_mask_quads": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "mpf_warp_composite": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mpf_warp_composite_split": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mpf_src_flow": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "mpf_src_flow_hard": (c_i, [c_p, c_i64, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "mpf_warp_composite_views": (c_i, [c_p, c_i, ctypes.POINTER(MpfWarpView), c_i, c_i, c_i, c_i, c_p]),
    "mpf_warp_views_and_blend_next": (c_i, [c_p, ctypes.POINTER(MpfWarpView), c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "mpf_warp_views_blend_next_merge_prev": (c_i, [c_p, ctypes.POINTER(MpfWarpView), c_i, c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i,
                                                  ctypes.POINTER(MpfMergeArgs), c_p]),
    "mpf_merge": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_p, c_p, c_p, c_p]),
    "mpf_merge_ex": (c_i, [ctypes.POINTER(MpfMergeArgs), c_i, c_i, c_p]),
    "mpf_merge_depth_ordered": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_i, c_p, c_p, c_p]),
    "mpf_fill_holes_workspace": (c_sz, [c_i, c_i]),
    "mpf_fill_holes": (c_i, [c_p, c_p, c_i, c_i, c_p, c_sz, c_p]),
    "mpf_prepare_inputs": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "mpf_inpaint_host": (c_i, [c_p, c_p, c_i, c_i, c_i, ctypes.c_double, c_i, c_p]),
    "mpf_png_filter_up": (c_i, [c_p, c_i, c_i, c_p, c_p]),
    "mpf_pair_stats": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "mpf_stream_probe": (c_i, [c_p, c_p, ctypes.c_size_t, c_i, c_p]),
    "mpf_to_u8_bgr": (c_i, [c_p, c_i, c_i, c_p, c_p]),
    "mpf_src_xyz": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "mpf_transform_xyz": (c_i, [c_p, c_p, c_i, c_i64, c_p, c_p]),
    "mpf_homography_sample": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "mpf_homography_flow": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "mpf_volume_render": (c_i, [c_p, c_p, c_p, c_i, c_i64, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "mpf_weighted_sum": (c_i, [c_p, c_p, c_i, c_i, c_i64, c_p, c_p]),
    "mpf_alpha_composite": (c_i, [c_p, c_p, c_i, c_i, c_i64, c_p, c_p, c_p, c_p]),
    "mpf_disp_to_depth": (c_i, [c_p, c_i64, c_p, c_p]),
    "mpf_backproject_project": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    "mpf_backproject": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "mpf_project3d": (c_i, [c_p, c_p, c_f, c_i, c_i, c_p, c_p, c_p]),
    "mpf_select_truncate": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mpf_moving_object_project": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mpf_forward_warp_workspace": (c_sz, [c_i, c_i]),
    "mpf_forward_warp": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_sz, c_p]),
    "mpf_warp_masks": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mpf_moving_object_chain": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, ctypes.POINTER(MpfMovingObjectOut), c_p, c_sz, c_p]),
    "forward_warping": (None, [c_p, c_p, c_p, c_p, c_p, c_i, c_i]),
    "mpf_forward_warping_host": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i]),
    "mpf_tune": (c_i, [ctypes.c_char_p, c_i]),
    "mpf_is_witness_build": (c_i, []),
    "mpf_conv3x3_f16": (c_i, [ctypes.POINTER(MpfConvArgs), c_p]),
    "mpf_plane_masks": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_p), c_p]),
    "mpf_encoder_input": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "mpf_conv2d_f32": (c_i, [ctypes.POINTER(MpfConv2dArgs), c_p]),
    "mpf_maxpool3x3s2_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "mpf_pconv": (c_i, [ctypes.POINTER(MpfPConvArgs), c_p]),
    "mpf_pfmn_input": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "mpf_pencoder_input": (c_i, [c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "mpf_pbilinear2x": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "mpf_pper_plane": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "mpf_pplane_masks": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_p), c_i, c_p]),
    "mpf_pmaxpool3x3s2": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
}


class MpiFlowHipError(RuntimeError):
    pass


def _bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load_witness():
    """The witness build (libmpiflow_hip_witness.so), loaded once, NOT made the library the package's calls go through (see witness())."""
    global _witness
    if _witness is None:
        if not os.path.exists(WITNESS_PATH):
            raise MpiFlowHipError("libmpiflow_hip_witness.so not found at %s - `make -C mpiflow_amd/csrc` builds it beside the product library" % WITNESS_PATH)
        _witness = _bind(WITNESS_PATH)
        assert _witness.mpf_is_witness_build() == 1
    return _witness


class witness:
    """`with _lib.witness() as lib:` - inside the block every call of the package goes through the WITNESS build, whose mpf_tune accepts the variant / ablation
    keys ("stage_b", "planar_lds", "fwarp_path", "ovl_depth", "ovl_xcd_a", "view_shift", "ovl_ablate").  Tests and tools only; not re-entrant, one thread."""

    def __enter__(self):
        global _lib
        self.prev = load()
        _lib = load_witness()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def select_witness():
    """Process-wide switch to the witness build (tools/ that time retired variants; bench.py --witness)."""
    global _lib
    _lib = load_witness()
    return _lib


def load():
    """Load the HIP library once.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MpiFlowHipError(
            "libmpiflow_hip.so not found at %s - build it with `python __graft_entry__.py` or "
            "`make -C mpiflow_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    skipped = []
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("MPIFLOW_HIP_LIB") and not hasattr(lib, name):
            skipped.append(name)
            continue
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    if skipped:                          # development hook only: say which calls will fail instead of failing late with an AttributeError
        import sys
        sys.stderr.write("mpiflow_amd: MPIFLOW_HIP_LIB=%s lacks %d symbol(s) of include/mpiflow_hip.h: %s\n" % (LIB_PATH, len(skipped), ", ".join(skipped)))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mpf_last_error()
        raise MpiFlowHipError("%s failed with code %d: %s" % (what or "libmpiflow_hip call", rc,
                                                              msg.decode() if msg else "?"))


def device_info(device=0):
    lib = load()
    cu, mem = c_i(0), c_sz(0)
    arch = ctypes.create_string_buffer(64)
    check(lib.mpf_device_info(device, ctypes.byref(cu), ctypes.byref(mem), arch, 64), "mpf_device_info")
    return dict(cu_count=cu.value, hbm_bytes=mem.value, arch=arch.value.decode())
