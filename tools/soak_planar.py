#!/usr/bin/env python3
"""Randomised soak of Stage B on the reference's channel-planar tensors: random (S, H, W), poses from mild to extreme (footprints that do not
fit the LDS tile take the gather body per tile), with / without mask, with / without depth + validity outputs, one [S,4,H,W] tensor and split
rgb / sigma tensors - the LDS-staged kernel (mpf_tune("planar_lds", 1), the default) must equal the gather kernel (0) and the interleaved
pipeline kernel bit for bit.  usage: soak_planar.py [n_cases] [seed]"""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import _lib, host_math, ops, synth

dev = torch.device("cuda:0")
lib = _lib.select_witness()          # the variant keys this tool switches exist in the witness build only (libmpiflow_hip_witness.so)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0


def same(a, b):
    return (a is None and b is None) or torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


for case in range(n_cases):
    S, H, W = rng.choice([1, 2, 3, 7, 16, 31, 64]), rng.randint(1, 120), rng.randint(1, 200)
    g = torch.Generator(device=dev).manual_seed(1000 + case)
    stack = torch.rand((S, 4, H, W), generator=g, device=dev)
    stack[:, 3] = torch.relu(3.0 * torch.randn((S, H, W), generator=g, device=dev) - 3.0) + 1e-4
    rgb3, sig1 = stack[:, :3].contiguous().clone(), stack[:, 3:].contiguous().clone()
    inter = ops.alloc_rgba_stack(S, H, W, dev)
    inter.copy_(stack.permute(0, 2, 3, 1))
    K = synth.intrinsics(H, W)
    if rng.random() < 0.3:
        K = K.copy(); K[0, 1] = 0.02 * W                     # skewed intrinsics: the dense K^-1 path
    k_inv = host_math.k_inverse(K)
    d = host_math.plane_depths(synth.plane_disparities(S))
    scale = rng.choice([0.05, 0.15, 0.15, 0.6, 2.0])         # 0.6 / 2.0: far beyond the reference sampler - footprints that do not fit
    G = host_math.generate_random_pose(scale, rng=rng)
    if scale >= 0.6:
        G = G.clone(); G[:3, 3] *= 4.0
    H_ts, H_st = host_math.homographies(G, k_inv, K, d)
    om = torch.rand((H, W), generator=g, device=dev)
    for use_mask in (True, False):
        quads = ops.mask_quads(om) if use_mask else None
        for aux in (True, False):
            ref = ops.warp_composite(inter, quads, H_st, k_inv, G, d, interleaved=2, want_depth=aux, want_tgt_mask=aux)
            outs = {}
            for lds in (1, 0, 2):
                _lib.check(lib.mpf_tune(b"planar_lds", lds))
                outs[(lds, "stack")] = ops.warp_composite(stack, quads, H_st, k_inv, G, d, interleaved=False, want_depth=aux, want_tgt_mask=aux)
                outs[(lds, "split")] = ops.warp_composite_split(rgb3, sig1, quads, H_st, k_inv, G, d, want_depth=aux, want_tgt_mask=aux)
            _lib.check(lib.mpf_tune(b"planar_lds", 1))
            torch.cuda.synchronize()
            for key, o in outs.items():
                ok = all(same(o[k], ref[k]) for k in ("rgb", "depth", "tgt_mask", "objmask"))
                if not ok:
                    bad += 1
                    print("MISMATCH case %d: S=%d H=%d W=%d pose scale %s mask=%s aux=%s planar_lds=%d %s" % (case, S, H, W, scale, use_mask, aux, key[0], key[1]), flush=True)
print("soak planar: %d cases x 16 variants, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
