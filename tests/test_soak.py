"""Randomised soak of the pipelined pair stream (tools/soak_pipeline.py, a fixed seed and a bounded number of cases): random (S, H, W) incl.
one-pixel-wide frames, streams of 1-6 pairs with DIFFERENT inputs per pair through pipeline.OverlappedPairRenderer in all six modes (merge as a
launch / in the launch x no chain / ordered chain / independent chain): every pair's outputs equal pipeline.render_pair's, the chain's outputs
the stand-alone chain's, bit for bit.  (Found in round 4: with merge_in_launch a pair is handed back two push() calls later, so the chain's
output sets must be a ring of three - identical inputs for every pair had hidden it.)"""
import os
import subprocess
import sys

import pytest
import torch

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


def test_engine_first_layer_factorisation_soak():
    """tools/soak_engine_factor.py: the producer with its first layer synthesised in the consumers' loaders against the materialised form, random
    sizes / plane counts / parameters: equal to fp16-rounding level on logits, cumulative mask and output."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_engine_factor.py"), "6", "1"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]
