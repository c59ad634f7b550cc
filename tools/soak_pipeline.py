#!/usr/bin/env python3
"""Randomised soak of the pipelined pair stream: random (S, H, W), random poses / masks / stacks, streams of 1-6 pairs through
pipeline.OverlappedPairRenderer in every mode (merge launch / merge in launch; no chain / ordered chain / independent chain) - every pair's
outputs must equal, bit for bit, what pipeline.render_pair returns for it, and the chain's outputs what the stand-alone chain returns.
usage: soak_pipeline.py [n_cases] [seed]"""
import os, random, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpiflow_amd import host_math, moving_obj, pipeline, synth

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0


def same(a, b):
    """bit patterns, so that the NaNs a 1-pixel-wide frame produces (geometry.py divides by W - 1) compare equal to themselves"""
    if a.dtype.is_floating_point:
        a, b = a.contiguous().view(torch.int32), b.contiguous().view(torch.int32)
    return torch.equal(a, b)


for case in range(n_cases):
    S, H, W = rng.choice([1, 2, 3, 5, 8, 16, 17, 33]), rng.randint(1, 90), rng.randint(1, 150)
    n = rng.randint(1, 6)
    K, pd = synth.intrinsics(H, W), synth.plane_disparities(S)
    g = torch.Generator(device=dev).manual_seed(case)
    pairs = []
    for k in range(n):
        mpi = torch.rand((S, 4, H, W), generator=g, device=dev)
        mpi[:, 3] = torch.relu(3.0 * torch.randn((S, H, W), generator=g, device=dev) - 3.0) + 1e-4
        img = torch.rand((3, H, W), generator=g, device=dev)
        om = (torch.rand((H, W), generator=g, device=dev) > 0.6).float() * torch.rand((H, W), generator=g, device=dev).clamp(min=0.3)
        disp = torch.rand((H, W), generator=g, device=dev)
        Gc = host_math.generate_random_pose(0.15, base_motions=(0, 0, 0), rng=rng)
        Gd = host_math.generate_random_pose(0.15, rng=rng)
        pairs.append((mpi, img, om, disp, Gc, Gd))
    T_obj = host_math.transformation_from_parameters(torch.zeros(1, 1, 3), torch.tensor([[rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(0.0, 0.1)]]))
    inv_K = torch.inverse(torch.from_numpy(K).double()).float()
    ref = []
    ref_chain = moving_obj.MovingObjectChain(H, W, K, inv_K, dev, T_obj=T_obj, n_buffers=1)
    for mpi, img, om, disp, Gc, Gd in pairs:
        o = pipeline.render_pair(img, om, mpi, pd, K, Gc, Gd)
        b = ref_chain.run(disp, om, o["src_np"])
        ref.append(([o["flow_mix"].clone(), o["frame_mix"].clone(), o["fill_mask"].clone()],
                    [b.p1.clone(), b.z1.clone(), b.safe_x.clone(), b.safe_y.clone(), b.flow_01.clone(), b.warped.clone()] + [b.masks[k].clone() for k in sorted(b.masks)]))
    for mil in (False, True):
        for chain_mode in (None, "ordered", "independent"):
            ovl = pipeline.OverlappedPairRenderer(S, H, W, dev, merge_in_launch=mil)
            if chain_mode:
                ovl.attach_chain(moving_obj.MovingObjectChain(H, W, K, inv_K, dev, T_obj=T_obj, n_buffers=3), ordered=chain_mode == "ordered")
            got = []

            def take(d):
                if d is None:
                    return
                r = [t.clone() for t in d[:3]]
                c = None
                if chain_mode:
                    b = d[3]
                    if chain_mode == "independent":
                        b.ready.synchronize()
                    c = [b.p1.clone(), b.z1.clone(), b.safe_x.clone(), b.safe_y.clone(), b.flow_01.clone(), b.warped.clone()] + [b.masks[k].clone() for k in sorted(b.masks)]
                got.append((r, c))
            for mpi, img, om, disp, Gc, Gd in pairs:
                take(ovl.push(mpi, img, ovl.prepare(K, pd, [Gc, Gd]), om, moving=(disp, om) if chain_mode else None))
            for d in ovl.flush():
                take(d)
            torch.cuda.synchronize()
            ok = len(got) == n
            for (r, c), (rr, rc) in zip(got, ref):
                ok = ok and all(same(a, b) for a, b in zip(r, rr))
                if chain_mode:
                    ok = ok and all(same(a, b) for a, b in zip(c, rc))
            if not ok:
                bad += 1
                print("MISMATCH case %d: S=%d H=%d W=%d n=%d merge_in_launch=%s chain=%s" % (case, S, H, W, n, mil, chain_mode), flush=True)
print("soak: %d cases x 6 modes, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
