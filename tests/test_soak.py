"""Randomised soak of the pipelined pair stream (tools/soak_pipeline.py, a fixed seed and a bounded number of cases): random (S, H, W) incl.
one-pixel-wide frames, streams of 1-6 pairs with DIFFERENT inputs per pair through pipeline.OverlappedPairRenderer in all six modes (merge as a
launch / in the launch x no chain / ordered chain / independent chain): every pair's outputs equal pipeline.render_pair's, the chain's outputs
the stand-alone chain's, bit for bit.  (Found in round 4: with merge_in_launch a pair is handed back two push() calls later, so the chain's
output sets must be a ring of three - identical inputs for every pair had hidden it.)"""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import bits_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 99])
def test_pipeline_soak(seed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_pipeline.py"), "40", str(seed)], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_planar_stage_b_soak():
    """tools/soak_planar.py: Stage B on the reference's planar tensors, LDS-staged vs gather vs interleaved kernel, random shapes and poses from
    mild to extreme (per-tile fallback), with / without mask and depth outputs, one tensor and split tensors: bit-identical."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_planar.py"), "60", "3"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_forward_warp_paths_soak():
    """tools/soak_fwarp.py: the forward warp's three paths - the round-5 gather (default), the general radix path, round 2's sort + bucket workgroups - on
    random sizes from one pixel up with smooth flows, white noise in a box (bench.py's c3), regions clamped onto border pixels (pile-ups streamed through
    the gather kernel's LDS in chunks), everything onto a few targets, uniformly random targets, sentinel / tied / NaN z: same bytes, and the serial C
    restatement's on every fourth small case."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_fwarp.py"), "120", "5"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_engine_first_layer_factorisation_soak():
    """tools/soak_engine_factor.py: the producer with its first layer synthesised in the consumers' loaders against the materialised form, random
    sizes / plane counts / parameters: equal to fp16-rounding level on logits, cumulative mask and output."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_engine_factor.py"), "6", "1"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_engine_phase_decomposition_soak():
    """tools/soak_engine_phase.py: the phase-decomposed x2-nearest layers (round 6) against the gather form on random sizes / plane counts / weights - equal to
    fp16-rounding level; the phase form repeated, and with upconv(0,1) walking 1 / 2 / 4 planes per workgroup, bit-identical."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_engine_phase.py"), "6", "3"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_encoder_kernels_soak():
    """tools/soak_encoder.py: the fp32 single-image convolution / max-pool kernels on random shapes (kernel 1 / 3 / 7, strides, paddings, x2 nearest
    in front, residual, activations, partial tiles, both split-K variants) and the whole encoder at random sizes, against torch in fp64."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_encoder.py"), "40", "2"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_precise_engine_matrix_core_form_soak():
    """tools/soak_precise.py: the parity-grade engine's fp32 form on the bf16 matrix cores at random plane counts / sizes / parameters: the forward twice on
    one input is bit-identical (fixed sum order; an LDS hand-off or barrier race in k_pconv_x3 / k_pconv_x3_tile would show), finite, and within fp32 bars of
    the fp32-instruction form (other kernels, no LDS)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_precise.py"), "10", "4"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_pair_vs_oracle_soak(oracle):
    """The fused pair against the CPU oracle (kernel-exp mode: same IEEE op sequence) on random small problems: every plane count from 1 to 40,
    frames from one pixel to 60 x 90, random stacks / images / soft masks, poses from the reference sampler's range up to 6 x beyond it (planes
    behind the camera, samples leaving the frame), both Stage B launch forms and the pipelined renderer (merge in the launch) - every output
    bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mpiflow_amd import pipeline, synth
    dev = torch.device("cuda:0")
    rng = random.Random(2024)
    oracle.set_exp_mode(1)
    try:
        for case in range(24):
            S, H, W = rng.randint(1, 40), rng.randint(1, 60), rng.randint(1, 90)
            rs = np.random.RandomState(case)
            mpi = rs.rand(S, 4, H, W).astype(np.float32)
            mpi[:, 3] = np.maximum(3.0 * rs.randn(S, H, W) - 3.0, 0.0).astype(np.float32) + np.float32(1e-4)
            img = rs.rand(3, H, W).astype(np.float32)
            om = (rs.rand(H, W) * (rs.rand(H, W) > 0.4)).astype(np.float32)
            K, pd = synth.intrinsics(H, W), synth.plane_disparities(S)
            scale = rng.choice([0.15, 0.15, 0.4, 0.9])
            G_dyn, G_cam = oracle.random_pose(rng, scale), oracle.random_pose(rng, scale, base_motions=(0, 0, 0))
            want = oracle.render_pair(img, om, mpi, pd, K, G_cam, G_dyn)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
            for multi_view in (True, False):
                r = pipeline.PairRenderer(S, H, W, dev)
                r.multi_view = multi_view
                got = pipeline.render_pair(t(img), t(om), t(mpi), pd, K, G_cam, G_dyn, renderer=r)
                for k in ("flow_mix", "frame_mix", "fill_mask", "src_np"):
                    assert bits_equal(got[k].cpu().numpy(), want[k]) == 0, (case, S, H, W, scale, multi_view, k)
                assert bits_equal(got["view_cam"]["rgb"].cpu().numpy(), want["view_cam"]["rgb"]) == 0
                assert bits_equal(got["view_dyn"]["objmask"].cpu().numpy(), want["view_dyn"]["objmask"]) == 0
            ovl = pipeline.OverlappedPairRenderer(S, H, W, dev, merge_in_launch=True)
            prep = ovl.prepare(K, pd, [G_cam, G_dyn])
            res = [ovl.push(t(mpi), t(img), prep, t(om)) for _ in range(3)]
            res = [x for x in res if x is not None] + ovl.flush()
            assert len(res) == 3
            for d in res:
                for k, a in zip(("flow_mix", "frame_mix", "fill_mask"), d):
                    assert bits_equal(a.cpu().numpy(), want[k]) == 0, (case, S, H, W, scale, "pipelined", k)
    finally:
        oracle.set_exp_mode(0)
