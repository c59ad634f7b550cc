"""CPU-only checks of the C-ABI boundary: the in-tree library loads (cross-compiled for gfx950, no GPU needed to dlopen),
exports every symbol include/mpiflow_hip.h declares, the ctypes table covers them all, argument validation returns error
codes (never aborts), and the product never routes through the oracle."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from mpiflow_amd import _lib
    return _lib


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mpiflow_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mpf_\w+|forward_warping)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    names = declared_symbols()
    assert "forward_warping" in names and "mpf_warp_composite" in names and len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libmpiflow_hip.so does not export %s" % n


def test_ctypes_table_matches_header(built):
    assert sorted(built.SIGNATURES) == declared_symbols()
    lib = built.load()
    assert lib.mpf_version() == 601


def test_bad_arguments_return_error_codes(built):
    lib = built.load()
    rc = lib.mpf_warp_composite(None, 1, None, None, 4, 8, 8, None, None, None, None, None, None)
    assert rc == 10001 and b"null pointer" in lib.mpf_last_error()
    one = ctypes.c_void_p(256)
    rc = lib.mpf_src_blend_flow(one, one, one, 3, 4, 8, 8, 0.0, None, None, None, None, None, None, None, None, None, None)
    assert rc == 10001 and b"P must be" in lib.mpf_last_error()
    rc = lib.mpf_warp_composite(one, 1, None, one, 5000, 8, 8, one, None, None, None, None, None)
    assert rc == 10001 and b"bad shape" in lib.mpf_last_error()
    assert lib.mpf_forward_warp_workspace(640, 960) > 5 * 640 * 960 * 4
    # round 5's entry points validate before they launch anything (no GPU needed to be told so)
    assert lib.mpf_pconv(None, None) == 10001 and b"null argument block" in lib.mpf_last_error()
    assert lib.mpf_merge_ex(None, 4, 4, None) == 10001
    assert lib.mpf_src_flow_hard(one, 10, one, 1, 4, 8, 8, 0.0, one, None) == 10001 and b"plane_stride" in lib.mpf_last_error()
    assert lib.mpf_pbilinear2x(ctypes.c_void_p(260), 1, 2, 2, 4, one, 0, None) == 10001 and b"aligned" in lib.mpf_last_error()
    # tuning knobs of the PRODUCT library: scheduling only.  Every key that selects a retired kernel variant or a timing ablation is refused - it cannot
    # change a result of this process, whoever calls it - and unknown keys are errors
    assert lib.mpf_is_witness_build() == 0
    for key in (b"sbf_px", b"conv_pf", b"chain_grid", b"chain_prio"):
        assert lib.mpf_tune(key, 0) == 0
    assert lib.mpf_tune(b"conv_pf", 1) == 0 and lib.mpf_tune(b"fwarp_gate", -1) == 0
    for key in (b"ovl_ablate", b"stage_b", b"planar_lds", b"fwarp_path", b"ovl_xcd_a", b"view_shift", b"ovl_depth", b"chain_stop"):
        assert lib.mpf_tune(key, 1) == 10001 and b"unknown key" in lib.mpf_last_error(), key
    with pytest.raises(built.MpiFlowHipError):
        built.check(rc, "probe")


def test_witness_build_keeps_the_variants_and_the_product_has_none_of_their_kernels(built):
    """libmpiflow_hip_witness.so (-DMPF_WITNESS): the same C ABI + the variant / ablation keys; the product's dynamic symbol table holds none of the retired
    kernels (nm -D), the witness's holds them all."""
    import subprocess
    w = built.load_witness()
    assert w.mpf_is_witness_build() == 1 and w.mpf_version() == built.load().mpf_version()
    for n in declared_symbols():
        assert hasattr(w, n), "libmpiflow_hip_witness.so does not export %s" % n
    for key, v in ((b"stage_b", 20), (b"stage_b", 1), (b"planar_lds", 2), (b"planar_lds", 1), (b"fwarp_path", 2), (b"fwarp_path", 0), (b"ovl_xcd_a", 0),
                   (b"view_shift", 8), (b"ovl_depth", 4), (b"ovl_ablate", 0)):
        assert w.mpf_tune(key, v) == 0, key
    assert w.mpf_tune(b"chain_stop", 1) == 10001
    retired = ("k_warp_composite_dbg", "k_warp_composite_lds", "k_warp_composite_planar_wave", "k_fw_bucket", "k_fw_keys_hist", "k_mo_project_keys_hist")
    syms = {p: subprocess.run(["nm", "-D", "--defined-only", p], capture_output=True, text=True, check=True).stdout for p in (built.LIB_PATH, built.WITNESS_PATH)}
    for k in retired:
        assert k not in syms[built.LIB_PATH], "%s is compiled into the product library" % k
        assert k in syms[built.WITNESS_PATH], k
    assert "k_pair_overlap" in syms[built.LIB_PATH] and "k_fw_gather_resolve" in syms[built.LIB_PATH]
    # the witness is a context, not a mode: calls go back to the product library afterwards
    with built.witness() as lib:
        assert lib is w and built.load() is w
    assert built.load() is not w and built.load().mpf_is_witness_build() == 0


def test_missing_library_fails_loudly(monkeypatch, built):
    import importlib
    from mpiflow_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmpiflow_hip.so")
    with pytest.raises(_lib.MpiFlowHipError, match="no CPU fallback"):
        _lib.load()


def test_cpu_tensors_are_rejected_not_silently_computed(built):
    import torch
    from mpiflow_amd import ops
    with pytest.raises(built.MpiFlowHipError, match="no CPU path"):
        ops.to_u8_bgr(torch.zeros(3, 4, 4))


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|liboracle|mpi_oracle", re.M)
    for base, _, files in os.walk(os.path.join(ROOT, "mpiflow_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(base, f)).read()
                assert not pat.search(txt), "%s references the oracle" % os.path.join(base, f)
    for f in ("gen_3dphoto_dynamic.py",):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert not pat.search(open(p).read())


def test_no_wide_buffer_store_uses_a_register_soffset(built, tmp_path):
    """gfx950: a buffer_store_dwordx3/x4 whose soffset is an SGPR, followed at once by a VALU write of its data registers, stores the NEW
    values in lanes 12-15 of every 16 (hipcc's hazard recogniser exempts that form; DESIGN.md §5 - found by the every-pixel test at
    64 x 640 x 960, invisible at small sizes).  The kernels keep the plane offset in the VGPR offset; this pins it in the BUILT code
    objects, so a later edit (or compiler) that moves a uniform offset into soffset is caught on the CPU."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not in this image")
    so = tmp_path / "lib.so"
    shutil.copy(built.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", so.name], cwd=tmp_path, check=True, capture_output=True)
    objs = [p for p in tmp_path.iterdir() if "amdgcn" in p.name and p.stat().st_size > 0]
    assert objs, "no gfx950 code object found in the library"
    n_wide = 0
    for o in objs:
        text = subprocess.run([objdump, "-d", o.name], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"buffer_store_dwordx[34] (v\[\d+:\d+\]), (\w+), (s\[\d+:\d+\]), (\S+)", text):
            n_wide += 1
            assert not m.group(4).startswith("s"), "wide buffer store with a register soffset: %s" % m.group(0)
    assert n_wide > 100                                     # the Stage A+C role of the overlapped launch stores this way
