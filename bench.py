#!/usr/bin/env python3
"""bench.py - image-pairs/sec of the MPI render + flow hot path on MI355X, with roofline and CPU baseline.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 64 planes, 640 x 960, camera-only novel view.  One *step* = every rank renders
`--images` distinct image pairs whose plane stacks are already resident in HBM (synthetic data of the shape AdaMPI
emits; random poses are fixed per image).  Per pair, two launches:
    mpf_src_blend_flow   Stage A+C: blend source image into the stack, volume-rendered flow for the pose   (P = 1)
                         (+ fused: source frame as uint8 BGR, bilinear tap quads of the all-ones object mask - the
                          reference's camera-only call passes an all-ones mask through the same 8-channel warp)
    mpf_warp_composite   Stage B  : 64-plane homography warp + front-to-back composite  <- dominant / roofline kernel
                         (+ fused: rendered frame as uint8 BGR; rendered object mask)
Images are independent, so ranks share nothing; the only collective is the end-of-batch statistics all-reduce
(RCCL over xGMI under torchrun), issued once after the K timed steps, inside the timed region.  `value` = pairs rendered by all ranks / max-over-ranks wall time.

`roofline`: Stage B's algorithmic bytes (16*S*N) over its mean launch duration, measured with HIP
events recorded on the launch stream around every Stage B launch inside the timed region; peak 8.0 TB/s.
`cpu_baseline`: the CPU oracle (our plain-C restatement of the reference algorithm, OpenMP) timed on this host on a
bounded sample of the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required for RCCL between processes on this driver

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mpiflow_amd import _lib, host_math, ops, pipeline, synth  # noqa: E402

HBM_PEAK = 8.0e12   # MI355X HBM3E, bytes/s (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--images", type=int, default=8, help="resident image stacks (pairs per step) per GPU")
    p.add_argument("--planes", type=int, default=64)
    p.add_argument("--height", type=int, default=640)
    p.add_argument("--width", type=int, default=960)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-pairs", type=int, default=6, help="pairs the CPU oracle renders for cpu_baseline")
    p.add_argument("--sbf-px", type=int, default=0, help="tuning: pixels/thread of Stage A+C (0 = library default)")
    p.add_argument("--streams", type=int, default=1, help="HIP streams the pairs of a step are spread over (each with its own blended stack)")
    return p.parse_args()


def init_dist(n):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (CPU-less CI of the multi-process path on a 1-GPU box): MPIFLOW_DIST_BACKEND=gloo, MPIFLOW_FORCE_DEVICE=0
    backend = os.environ.get("MPIFLOW_DIST_BACKEND", "nccl")
    if "MPIFLOW_FORCE_DEVICE" in os.environ:
        local = int(os.environ["MPIFLOW_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local


def make_image(S, H, W, dev, seed):
    """Synthetic AdaMPI-like stack generated on the device (SURVEY §8(d) distribution)."""
    g = torch.Generator(device=dev).manual_seed(1000 + seed)
    mpi = torch.empty((S, 4, H, W), dtype=torch.float32, device=dev)
    mpi[:, :3] = torch.rand((S, 3, H, W), generator=g, device=dev)
    mpi[:, 3] = torch.relu(3.0 * torch.randn((S, H, W), generator=g, device=dev) - 4.0) + 1e-4
    img = torch.rand((3, H, W), generator=g, device=dev)
    return mpi, img


def cpu_baseline(S, H, W, pairs):
    """Time the oracle (checker, used here only as the reported CPU baseline) on `pairs` camera-only pairs."""
    from oracle import mpi_oracle as orc
    inp = synth.make_inputs(S, H, W, seed=77, kind="white")
    aa, tr = synth.bench_pose()
    G = host_math.transformation_from_parameters(torch.tensor([[aa]], dtype=torch.float32), torch.tensor([tr], dtype=torch.float32))[0].numpy()
    d = orc.plane_depths(inp["disparity"])
    k_inv = orc.k_inverse(inp["K"])
    H_ts, H_st = orc.homographies(G, k_inv, inp["K"], d)
    ones = np.ones((H, W), np.float32)

    def one():
        a = orc.src_blend_flow(inp["mpi"], inp["image"], k_inv, d, H_ts[None])
        v = orc.warp_composite(a["rgba"], ones, H_st, k_inv, G, d)
        orc.to_u8_bgr(v["rgb"])
        orc.to_u8_bgr(inp["image"])

    one()
    t0 = time.perf_counter()
    for _ in range(pairs):
        one()
    dt = time.perf_counter() - t0
    return dict(value=pairs / dt, unit="pairs/s", cores=os.cpu_count(), kind="port",
                sample="%d camera-only pairs at %dx%dx%d by the plain-C oracle (OpenMP, %d threads), %.1f s" %
                       (pairs, S, H, W, os.cpu_count(), dt))


def main():
    a = parse()
    rank, world, local = init_dist(a.gpus)
    dev = torch.device("cuda", local)
    _lib.load()
    if a.sbf_px:
        _lib.check(_lib.load().mpf_tune(b"sbf_px", a.sbf_px))
    S, H, W, B = a.planes, a.height, a.width, a.images
    N = H * W
    K = synth.intrinsics(H, W)
    disp = synth.plane_disparities(S)

    # resident inputs + per-image fixed random pose (reference sampler, seed 114514, camera pose stream)
    import random
    rng = random.Random(114514 + rank)
    images, preps = [], []
    NS = max(1, a.streams)
    renderers = [pipeline.PairRenderer(S, H, W, dev, n_views=1) for _ in range(NS)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)] if NS > 1 else [torch.cuda.current_stream()]
    for i in range(B):
        images.append(make_image(S, H, W, dev, seed=rank * 1000 + i))
        G = host_math.generate_random_pose(0.15, rng=rng)
        preps.append(renderers[0].prepare(K, disp, [G]))
    ones = torch.ones((H, W), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    ev = []

    def step(timed):
        main = torch.cuda.current_stream()
        if NS > 1:
            for s_ in streams:
                s_.wait_stream(main)
        for i, ((mpi, img), prep) in enumerate(zip(images, preps)):
            r = renderers[i % NS]
            with torch.cuda.stream(streams[i % NS]):
                ops.src_blend_flow(mpi, img, out_rgba=r.rgba, out_flows=r.flows[:1], dparams=prep["blend"], P=1,
                                   src_u8=r.src_u8, obj_mask=ones, quads=r.quads[0])
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                ops.warp_composite(r.rgba, r.quads[0], dparams=prep["warp"][0], out=r.views[0], interleaved=2)
                if timed:
                    e1.record()
                    ev.append((e0, e1))
        if NS > 1:
            for s_ in streams:
                main.wait_stream(s_)
        return B

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(a.warmup):
        step(False)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = pipeline.empty_stats()
    for _ in range(a.steps):
        st["pairs"] += step(True)
    # end-of-batch statistics: the ONE collective of the path (SUM / MAX all-reduce of a 7-float vector, RCCL over xGMI)
    total_pairs = int(pipeline.reduce_stats(st)["pairs"])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    else:
        total_pairs = B * a.steps

    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) if ev else float("nan")
    alg_bytes = 16.0 * S * N
    achieved = alg_bytes / (kern_ms * 1e-3)
    traffic = None
    tp = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get("stage_b_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "image-pairs/sec (+flow) at 640x960x64 planes",
            "value": total_pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d planes, %dx%d, camera-only novel view (blend+flow, warp+composite, u8 frames)" % (S, H, W),
                       "pairs_per_step_per_gpu": B, "sharding": "independent images per rank, stats all-reduce only",
                       "device": _lib.device_info(local)},
            "roofline": {"bound": "hbm", "kernel": "k_warp_composite (Stage B)", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms, "launches_timed": len(ev)},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, H, W, a.cpu_pairs)
            out["cpu_baseline"]["reference_measured_in_build_container"] = \
                "reference render_3dphoto_dynamic (full dynamic pair) 64x640x960: 104.8 s on 8 threads (tests/golden/make_golden.py)"
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
