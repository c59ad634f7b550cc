#!/usr/bin/env python3
"""gen_3dphoto_dynamic.py - optical-flow training-pair generation from single images on MI355X.

Drop-in for the reference's entry point (gen_3dphoto_dynamic_v2.py; the README and scripts call it
gen_3dphoto_dynamic.py): same flags (--width --height --seed --ext_cz --ckpt_path --repeat --base --out), same input
layout (base/{images,disps,masks}), same outputs (out/{src_images,dst_images}/NAME_r.png, out/flows/NAME_r.flo), same
RNG draw order (np.random for the instance id, `random` for the two poses of every pair) - so a seeded run draws the
reference's instance ids and poses.

What is different:
  * the render/flow path runs in the HIP kernels of mpiflow_amd (fp32);
  * images are sharded over ranks when launched under torchrun (rank r takes images i = r mod world); every rank
    replays the whole RNG schedule, so an N-GPU run produces exactly the files of a 1-GPU run; one all-reduce of a
    7-float statistics vector (RCCL over xGMI) closes the batch;
  * the MPI producer: `--mpi-from model` runs the AdaMPI network (mpiflow_amd.model, state-dict compatible with the
    reference's checkpoints: `--ckpt_path adampi_64p.pth`, or `--ckpt_path random:SEED` for deterministic random weights -
    the published weights are not in the reference tree); its raw last-layer output is handed to Stage A+C, which applies
    the activation epilogue in registers.  `--mpi-from npz` reads precomputed stacks from base/mpis/NAME.npz (arrays `mpi`
    [S,4,H,W], `disparity` [S]); `--mpi-from disparity` (default) builds a hard-assignment MPI from the monocular disparity
    map: every plane carries the image colours, the plane nearest to the pixel's disparity is opaque.  All three feed the
    identical render path.
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mpiflow_amd import _lib, host_math, io_formats, ops, pipeline, synth  # noqa: E402
from mpiflow_amd.utils import utils as U  # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--height", type=int, default=384)
    p.add_argument("--seed", type=int, default=114514)
    p.add_argument("--ext_cz", type=float, default=0.15)
    p.add_argument("--ckpt_path", type=str, default="adampiweight/adampi_64p.pth", help="accepted for CLI parity; unused")
    p.add_argument("--repeat", type=int, default=5)
    p.add_argument("--base", type=str, required=True)
    p.add_argument("--out", type=str, required=True)
    p.add_argument("--poses", choices=["v2", "coco", "copy"], default="v2",
                   help="pose sampler constants: utils/utils.py (gen_3dphoto_dynamic_v2.py), utils/utils_coco.py or 'utils/utils copy.py'")
    p.add_argument("--planes", type=int, default=64)
    p.add_argument("--mpi-from", choices=["disparity", "npz", "model"], default="disparity")
    p.add_argument("--model-dtype", choices=["fp32", "fp16", "bf16"], default="fp32", help="autocast dtype of the network's convolutions")
    p.add_argument("--model-engine", choices=["torch", "hip"], default="torch",
                   help="torch: every convolution on PyTorch/MIOpen (fp32 = the reference's CPU numerics); hip: the per-plane networks on "
                        "the MFMA convolution engine (fp16 storage, fp32 accumulate - the reference's GPU precision), one hipGraph per image")
    p.add_argument("--inpaint", choices=["auto", "cv2", "hip", "none"], default="auto")
    p.add_argument("--writers", type=int, default=8, help="writer threads (PNG encode + file I/O overlap the GPU); 0 = synchronous")
    p.add_argument("--lanes", type=int, default=1,
                   help="images in flight on this GPU, each with its own streams, plane-stack buffer and network graph (same files for any "
                        "value).  Measured on MI355X: 1 lane 359 pairs/s, 2 lanes 267, 3 lanes 298 - the kernels are sized to fill the GPU on "
                        "their own, concurrent images only fight over the caches")
    opt, _ = p.parse_known_args(argv)
    return opt


def mpi_from_disparity(image_3HW, disp_HW, S):
    """Stand-in MPI producer: colours on every plane, sigma = 1e-4 except 50 on the plane nearest to the pixel's
    disparity (a hard depth assignment).  Returns (mpi [S,4,H,W], disparity [S])."""
    planes = torch.from_numpy(synth.plane_disparities(S)).to(disp_HW.device)
    idx = (disp_HW.unsqueeze(0) - planes.view(S, 1, 1)).abs().argmin(0)
    sigma = torch.full((S,) + tuple(disp_HW.shape), 1e-4, dtype=torch.float32, device=disp_HW.device)
    sigma.scatter_(0, idx.unsqueeze(0), 50.0)
    mpi = torch.cat([image_3HW.unsqueeze(0).expand(S, -1, -1, -1), sigma.unsqueeze(1)], dim=1).contiguous()
    return mpi, planes


def main(argv=None):
    opt = parse(argv)
    print(opt)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (multi-process path on a 1-GPU box): MPIFLOW_DIST_BACKEND=gloo, MPIFLOW_FORCE_DEVICE=0
    backend = os.environ.get("MPIFLOW_DIST_BACKEND", "nccl")
    if "MPIFLOW_FORCE_DEVICE" in os.environ:
        local = int(os.environ["MPIFLOW_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)        # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)

    random.seed(opt.seed)                         # gen_3dphoto_dynamic_v2.py:38-39
    np.random.seed(opt.seed)
    K = torch.tensor([[0.58, 0, 0.5], [0, 0.58, 0.5], [0, 0, 1]])      # :42-49
    K[0, :] *= opt.width
    K[1, :] *= opt.height
    K = K.unsqueeze(0)

    out = opt.out
    if rank == 0:
        for d in ("", "src_images", "dst_images", "flows", "obj_mask"):
            os.makedirs(os.path.join(out, d), exist_ok=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()

    img_base, disp_base, mask_base = (os.path.join(opt.base, d) for d in ("images", "disps", "masks"))
    names = sorted(os.listdir(img_base))
    model = hip_model = None
    amp = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(opt.model_dtype)
    if opt.mpi_from == "model":
        from mpiflow_amd.model import MPIPredictor
        if opt.ckpt_path.startswith("random:"):
            model = MPIPredictor(opt.width, opt.height, opt.planes).randomize_(int(opt.ckpt_path.split(":")[1])).eval().to(dev)
        else:
            model = MPIPredictor.from_checkpoint(opt.ckpt_path, opt.width, opt.height).to(dev)      # :52-60
            opt.planes = model.num_planes
    use_hip_model = model is not None and opt.model_engine == "hip"

    class Lane:
        """Everything one in-flight image owns: its streams, the blended plane stack, the network graph and its static buffers."""

        def __init__(self):
            self.stream, self.tail_stream, self.tail_ready = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Event()
            self.renderer = pipeline.PairRenderer(opt.planes, opt.height, opt.width, dev)
            self.fill_ws = torch.empty(int(_lib.load().mpf_fill_holes_workspace(opt.height, opt.width)), dtype=torch.uint8, device=dev)
            self.hip_model = None
            if use_hip_model:
                from mpiflow_amd.model.engine import HipPredictor
                self.hip_model = HipPredictor(model, encoder_dtype=amp, graph=True)

    lanes = [Lane() for _ in range(max(1, opt.lanes))]
    dstats = pipeline.DeviceStats(dev)
    ring = io_formats.OutputRing(opt.height, opt.width, dev, slots=max(4, 2 * max(opt.writers, 1)), threads=max(opt.writers, 1))
    t_start = time.perf_counter()
    prof = {}

    class lap:                                   # MPIFLOW_PROFILE=1: cumulative host seconds per stage (with a device sync per lap)
        on = bool(os.environ.get("MPIFLOW_PROFILE"))

        def __init__(self, key):
            self.key = key

        def __enter__(self):
            self.t = time.perf_counter()

        def __exit__(self, *exc):
            if lap.on:
                if os.environ.get("MPIFLOW_PROFILE") != "host":          # "host": submission time only, no device sync
                    torch.cuda.synchronize()
                prof[self.key] = prof.get(self.key, 0.0) + time.perf_counter() - self.t

    import torch.nn.functional as F
    inputs = io_formats.InputPrefetcher(names, img_base, disp_base, mask_base, owned=lambda i: (i % world) == rank, pin=True)
    n_pairs, t_first, n_first, n_owned = 0, None, 0, 0
    it = iter(inputs)
    while True:
        with lap("wait for decoded inputs"):
            item = next(it, None)
        if item is None:
            break
        i, img, mask_max, ids_host, image_host, disp_host = item
        name = img.split(".")[0]
        mine = image_host is not None
        lane = lanes[n_owned % len(lanes)]
        n_owned += int(mine)
        renderer, hip_model, tail_stream, tail_ready, fill_ws = lane.renderer, lane.hip_model, lane.tail_stream, lane.tail_ready, lane.fill_ws
        lane_ctx = torch.cuda.stream(lane.stream)
        lane_ctx.__enter__()                                      # everything this image enqueues goes to its lane's stream
        if mine:
            with lap("upload + resize image, disparity, mask"):
                image = image_host.to(dev, non_blocking=True)[None]
                disp = disp_host.to(dev, non_blocking=True)[None]
                ids = ids_host.to(dev, non_blocking=True)
                image = F.interpolate(image, size=(opt.height, opt.width), mode="bilinear", align_corners=True)   # :86-89
                disp = F.interpolate(disp, size=(opt.height, opt.width), mode="bilinear", align_corners=True)
            cum_mask = None
            with lap("MPI producer + blend"):
                if opt.mpi_from == "npz":
                    z = np.load(os.path.join(opt.base, "mpis", name + ".npz"))
                    mpi, planes = torch.from_numpy(z["mpi"]).to(dev), torch.from_numpy(z["disparity"]).to(dev)
                elif hip_model is not None:
                    mpi, cum_mask, planes = hip_model(image, disp)             # static buffers: consumed by blend() below
                elif model is not None:
                    with torch.no_grad(), torch.autocast("cuda", dtype=amp, enabled=amp is not None):      # :92-93
                        raw, cm, pd = model(image, disp, raw=True)
                    mpi, cum_mask, planes = raw[0].float().contiguous(), cm[0].float().contiguous(), pd[0].float()
                else:
                    mpi, planes = mpi_from_disparity(image[0], disp[0, 0], opt.planes)
                renderer.blend(mpi, image[0], K, planes, cum_mask=cum_mask)      # once per image; the `repeat` pairs below reuse it
                ring.submit_source(ops.png_scanlines(renderer.src_u8), [os.path.join(out, "src_images", f"{name}_{r}.png") for r in range(opt.repeat)])  # :122
        # every rank draws for every pair, so the stream position is identical to a single-process run.  The reference interleaves
        # an np.random draw (instance id, :101) with 24 `random` draws (two poses, utils.py:207-208) per pair; the two generators
        # are independent, so the image's draws are taken in that order here and its 2 x repeat poses built in one batched call
        with lap("pose draws"):
            obj_indices, pose_params = [], []
            for r in range(opt.repeat):
                obj_indices.append(np.random.randint(mask_max) + 1)
                pose_params.append(host_math.draw_pose_parameters(opt.ext_cz, profile=opt.poses))
                pose_params.append(host_math.draw_pose_parameters(opt.ext_cz, base_motions=[0, 0, 0], profile=opt.poses))
            poses = host_math.poses_from_parameters(pose_params) if mine else None
        for r in range(opt.repeat):
            if not mine:
                continue
            obj_index, cam_ext_dynamic, cam_ext = obj_indices[r], poses[2 * r], poses[2 * r + 1]
            with lap("instance mask"):
                obj_mask = (ids == obj_index).to(torch.float32)[None, None]                                       # :102-105
                obj_mask = F.interpolate(obj_mask, size=(opt.height, opt.width), mode="bilinear", align_corners=True)
            with lap("render pair"):
                res = pipeline.render_pair(image[0], obj_mask[0, 0], mpi, planes, K, cam_ext, cam_ext_dynamic, renderer=renderer,
                                           cum_mask=cum_mask, reuse_blend=True)
            # the tail of a pair (hole fill: one workgroup; scanlines; statistics; copies to the host) runs on a second stream, so
            # it overlaps the next pair's render instead of serialising a one-CU kernel into the main stream
            tail_ready.record()
            for tns in (res["frame_mix"], res["fill_mask"], res["flow_mix"]):
                tns.record_stream(tail_stream)
            with torch.cuda.stream(tail_stream):
                tail_stream.wait_event(tail_ready)
                with lap("hole fill + PNG scanlines"):
                    if opt.inpaint == "hip" or (opt.inpaint == "auto" and not U.have_cv2()):
                        frame = ops.fill_holes(res["frame_mix"], res["fill_mask"], workspace=fill_ws)
                        scan = ops.png_scanlines(frame)
                    else:                                                                                         # :284-286 on the host
                        frame = U._inpaint(res["frame_mix"], res["fill_mask"], opt.inpaint)
                        scan = torch.from_numpy(io_formats.filter_up_rgb(np.asarray(frame)[:, :, ::-1]))
                with lap("statistics + hand-off to the writers"):
                    dstats.add(res["flow_mix"], res["fill_mask"])
                    ring.submit_pair(res["flow_mix"], scan, os.path.join(out, "flows", f"{name}_{r}.flo"),            # :120
                                     os.path.join(out, "dst_images", f"{name}_{r}.png"))                              # :121
            n_pairs += 1
        lane_ctx.__exit__(None, None, None)
        if mine and t_first is None and n_owned >= len(lanes):
            t_first, n_first = time.perf_counter(), n_pairs       # start-up (graph capture, first MIOpen calls) ends once every lane has run
    with lap("drain writers"):
        torch.cuda.synchronize()
        ring.close()
    if lap.on and rank == 0:
        for k, v in prof.items():
            print("  %-40s %8.3f s" % (k, v))
    stats = dstats.result(n_pairs)
    t_end = time.perf_counter()
    stats["wall_seconds"] = t_end - t_start
    if t_first is not None and n_pairs > n_first and rank == 0:
        print("steady state after the first image: %.1f pairs/s on this rank" % ((n_pairs - n_first) / (t_end - t_first)))
    total = pipeline.reduce_stats(stats)
    if rank == 0:
        print("pairs %d  mean|flow| %.3f px  max|flow| %.2f px  hole px/pair %.0f  wall %.1f s  (%d rank%s)" % (
            total["pairs"], total["sum_flow_mag"] / max(total["pairs"], 1) / (opt.height * opt.width), total["max_flow_mag"],
            total["hole_px"] / max(total["pairs"], 1), total["wall_seconds"], world, "s" if world > 1 else ""))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
