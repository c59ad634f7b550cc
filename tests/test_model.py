"""AdaMPI producer network: state-dict compatibility with the reference and output parity.

The golden was produced by the REFERENCE model/AdaMPI.py loaded (strict) with MPIPredictor.randomize_(seed) parameters
(tests/golden/make_golden.py::gen_model).  Real weights are not available offline, so parity is on deterministic random
parameters; the dense convolutions are stock PyTorch (MKL-DNN on CPU, MIOpen on ROCm)."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs


def _sha_keys(sd):
    return hashlib.sha256(np.frombuffer("|".join(sorted(sd)).encode(), np.uint8).tobytes()).hexdigest()


def test_state_dict_keys_match_reference_checkpoints():
    from mpiflow_amd.model import MPIPredictor
    g = load_golden("model_adampi")
    m = MPIPredictor(int(g["W"]), int(g["H"]), int(g["S"]))
    assert len(m.state_dict()) == int(g["n_state"])
    assert _sha_keys(m.state_dict()) == str(g["sha_keys"])
    assert "decoder.convs.(-'-u-p-c-o-n-v-'-,- -4-,- -0-).gated_conv.conv2d.weight" in m.state_dict()


def test_forward_matches_reference_on_cpu():
    from mpiflow_amd.model import MPIPredictor
    g = load_golden("model_adampi")
    m = MPIPredictor(int(g["W"]), int(g["H"]), int(g["S"])).randomize_(int(g["seed"])).eval()
    with torch.no_grad():
        mpi, disp = m(torch.from_numpy(g["image"]), torch.from_numpy(g["disp"]))
        raw, cum_mask, _ = m(torch.from_numpy(g["image"]), torch.from_numpy(g["disp"]), raw=True)
    assert tuple(mpi.shape) == (1, int(g["S"]), 4, int(g["H"]), int(g["W"]))
    assert np.array_equal(disp.numpy(), g["plane_disp"])
    assert max_abs(mpi[0, :, :, ::2, ::2].numpy(), g["mpi_sub"]) < 2e-5
    assert float(mpi[:, :, 3].min()) >= float(np.float32(1e-4)) and float(mpi[:, :, :3].min()) >= 0 and float(mpi[:, :, :3].max()) <= 1
    # the raw hand-off reproduces the activated stack: rgb = sigmoid(raw), sigma = relu(raw * cum_mask) + 1e-4
    re = torch.cat((torch.sigmoid(raw[:, :, :3]), torch.relu(raw[:, :, 3:] * cum_mask.unsqueeze(2)) + 1e-4), dim=2)
    assert torch.equal(re, mpi)


@pytest.mark.gpu
def test_forward_on_gpu_and_fused_epilogue():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mpiflow_amd import host_math, ops
    from mpiflow_amd.model import MPIPredictor
    g = load_golden("model_adampi")
    S, H, W = int(g["S"]), int(g["H"]), int(g["W"])
    dev = torch.device("cuda:0")
    m = MPIPredictor(W, H, S).randomize_(int(g["seed"])).eval().to(dev)
    img, dsp = torch.from_numpy(g["image"]).to(dev), torch.from_numpy(g["disp"]).to(dev)
    with torch.no_grad():
        mpi, disp = m(img, dsp)
        raw, cum_mask, _ = m(img, dsp, raw=True)
    assert max_abs(mpi[0, :, :, ::2, ::2].cpu().numpy(), g["mpi_sub"]) < 2e-3      # MIOpen vs MKL-DNN convolutions
    # Stage A+C fed with the raw decoder output + cum_mask (fused activation epilogue) == fed with the activated stack
    from mpiflow_amd import synth
    K = synth.intrinsics(H, W)
    k_inv = host_math.k_inverse(K)
    d = host_math.plane_depths(disp[0])
    G = host_math.generate_random_pose(0.15, rng=__import__("random").Random(2))
    H_ts, _ = host_math.homographies(G, k_inv, K, d)
    # (activate the SAME raw tensor with torch: two forward passes of MIOpen convolutions need not agree bit for bit)
    act = torch.cat((torch.sigmoid(raw[:, :, :3]), torch.relu(raw[:, :, 3:] * cum_mask.unsqueeze(2)) + 1e-4), dim=2)
    a = ops.src_blend_flow(act[0].contiguous(), img[0], k_inv, d, H_ts[None])
    b = ops.src_blend_flow(raw[0].contiguous(), img[0], k_inv, d, H_ts[None], cum_mask=cum_mask[0].contiguous())
    assert torch.equal(a["rgba"][..., 3], b["rgba"][..., 3])                  # sigma: mul, max, add - identical ops
    assert float((a["rgba"] - b["rgba"]).abs().max()) < 1e-6                  # rgb: sigmoid within a few ulp
    assert float((a["flows"] - b["flows"]).abs().max()) < 1e-4
