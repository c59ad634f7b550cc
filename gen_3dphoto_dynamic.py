#!/usr/bin/env python3
"""gen_3dphoto_dynamic.py - optical-flow training-pair generation from single images on MI355X.

Drop-in for the reference's entry point (gen_3dphoto_dynamic_v2.py; the README and scripts call it
gen_3dphoto_dynamic.py): same flags (--width --height --seed --ext_cz --ckpt_path --repeat --base --out), same input
layout (base/{images,disps,masks}), same outputs (out/{src_images,dst_images}/NAME_r.png, out/flows/NAME_r.flo), same
RNG draw order (np.random for the instance id, `random` for the two poses of every pair) - so a seeded run draws the
reference's instance ids and poses.

What is different:
  * the render/flow path runs in the HIP kernels of mpiflow_amd (fp32);
  * images are sharded over ranks - `--gpus N` starts one process per GPU itself, or launch under torchrun - (rank r takes images
    i = r mod world); every rank
    replays the whole RNG schedule, so an N-GPU run produces exactly the files of a 1-GPU run; one all-reduce of a
    7-float statistics vector (RCCL over xGMI) closes the batch;
  * the MPI producer: `--mpi-from model` runs the AdaMPI network (mpiflow_amd.model, state-dict compatible with the
    reference's checkpoints: `--ckpt_path adampi_64p.pth`, or `--ckpt_path random:SEED` for deterministic random weights -
    the published weights are not in the reference tree); its raw last-layer output is handed to Stage A+C, which applies
    the activation epilogue in registers.  `--mpi-from npz` reads precomputed stacks from base/mpis/NAME.npz (arrays `mpi`
    [S,4,H,W], `disparity` [S]); `--mpi-from disparity` builds a hard-assignment stand-in MPI from the monocular disparity
    map (every plane carries the image colours, the plane nearest to the pixel's disparity is opaque) - for smoke runs only.
    All three feed the identical render path.  The default is `model`, as in the reference; a missing checkpoint is an error.
  * an image that cannot be processed (undecodable file, mask without instances - where the reference dies with
    np.random.randint(0) -, a render error) is skipped and listed in out/skipped.txt instead of ending the batch; `--resume`
    skips images whose outputs already exist.  Neither changes the poses / instance ids any other image gets.
  * hole filling: cv2.inpaint when OpenCV is installed, else the same algorithm restated in libmpiflow_hip.so (host, on the
    writer threads, overlapped with the GPU).
"""
import argparse
import os
import random
import sys
import time

_T_PROCESS = time.perf_counter()                  # start-up is reported from here: it includes `import torch`

# dmabuf IPC (the only mode this node pool's driver supports) for RCCL's peer-memory exchange between the per-GPU processes;
# already exported by the launch environment, set here for a bare `torchrun gen_3dphoto_dynamic.py`
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    p.add_argument("--ckpt_path", type=str, default="adampiweight/adampi_64p.pth",
                   help="AdaMPI checkpoint {'num_planes', 'weight'} as the reference loads it (gen_3dphoto_dynamic_v2.py:52-58), or "
                        "random:SEED for deterministic random weights (the published weights are not redistributable here)")
    p.add_argument("--repeat", type=int, default=5)
    p.add_argument("--base", type=str, required=True)
    p.add_argument("--out", type=str, required=True)
    p.add_argument("--poses", choices=["v2", "coco", "copy"], default="v2",
                   help="pose sampler constants: utils/utils.py (gen_3dphoto_dynamic_v2.py), utils/utils_coco.py or 'utils/utils copy.py'")
    p.add_argument("--planes", type=int, default=64, help="plane count for --mpi-from disparity / random weights (a checkpoint carries its own)")
    p.add_argument("--mpi-from", choices=["model", "npz", "disparity"], default="model",
                   help="model: the AdaMPI network from --ckpt_path, as the reference always does; npz: precomputed stacks base/mpis/NAME.npz; "
                        "disparity: a hard-assignment stand-in built from the disparity map (NOT the reference's producer - for smoke runs)")
    p.add_argument("--model-dtype", choices=["auto", "fp16", "bf16", "fp32", "fp32-mfma", "fp64"], default="auto",
                   help="arithmetic of the network.  --model-engine hip: auto | fp16 = the fast engine (fp16 storage, fp32 accumulate, 7.7 ms per image); fp32 = the "
                        "PARITY-GRADE engine, every convolution on mpf_pconv in the arithmetic class of the reference's CPU path: fp32 tensors, every product from the "
                        "three bf16 pieces each fp32 factor is exactly the sum of (on the bf16 matrix cores; 2^-24 relative per product), fp32 accumulation in blocks of "
                        "64 products carried in fp64 - closer to the network in exact arithmetic than torch's own fp32; fp32-mfma = the same engine with the products on "
                        "v_mfma_f32_16x16x4_f32 (a third slower); fp64 = the same kernels in double (equals the torch modules run in double to 1e-10).  "
                        "--model-engine torch: auto | fp32 = plain fp32, fp16 | bf16 = torch autocast")
    p.add_argument("--model-engine", choices=["hip", "torch"], default="hip",
                   help="hip (default): the whole network on this repo's HIP kernels - the per-plane feature-mask UNet and gated decoder (> 98 %% of "
                        "the flops) on the fp16 MFMA convolution engine (fp16 storage, fp32 accumulate: the precision of the reference's own GPU run, which "
                        "calls .half() on model and inputs, gen_3dphoto_dynamic_v2.py:46,59,82-84), the single-image encoder / bottleneck in fp32 - one "
                        "hipGraph per image, 7.8 ms per 64 x 384 x 1280 image; torch: every "
                        "convolution on PyTorch/MIOpen (fp32 with --model-dtype fp32 = the reference's CPU numerics, 115 ms per image)")
    p.add_argument("--inpaint", choices=list(U.INPAINT_METHODS), default="auto",
                   help="hole filling of the rendered frame (reference: cv2.inpaint NS radius 3).  auto = cv2 when OpenCV is installed, else "
                        "builtin = the same algorithm restated in libmpiflow_hip.so, run on the writer threads; peel (alias hip) = the onion-peel "
                        "GPU kernel, NOT OpenCV's algorithm; none = leave holes white")
    p.add_argument("--resume", action="store_true", help="skip images whose outputs (all --repeat pairs) already exist; the RNG schedule is unaffected")
    p.add_argument("--gpus", type=int, default=0,
                   help="GPUs of this node to shard the images over, one process per GPU (the reference's model: scripts/gen_train_kitti15_v2.sh "
                        "starts one process per CUDA_VISIBLE_DEVICES).  0 = whatever the launcher says (WORLD_SIZE; 1 when run bare).  N > 1 from "
                        "a bare `python gen_3dphoto_dynamic.py` starts the N ranks itself (torch.distributed.run, RCCL over xGMI); under torchrun it "
                        "must equal the launcher's rank count")
    p.add_argument("--writers", type=int, default=0,
                   help="writer threads PER RANK (hole fill, PNG encode and file I/O overlap the GPU); 0 = the rank's share of the node's logical CPUs minus 6 (cores / ranks on this node - 6), "
                        "between 2 and 32 - the ranks of a node share its host cores")
    p.add_argument("--lanes", type=int, default=1,
                   help="images in flight on this GPU, each with its own streams, plane-stack buffer and network graph (same files for any "
                        "value).  Measured on MI355X (round 4, 64 images x 5 pairs, same box): 1 lane 417 - 426 pairs/s, 2 lanes 419 - 426, 3 lanes "
                        "415 - every kernel is sized to fill the GPU on its own, a second image in flight finds nothing left to use")
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


def outputs_exist(out, name, repeat):
    """--resume: all three files of every pair of an image are there and non-empty"""
    for r in range(repeat):
        for sub, ext in (("flows", "flo"), ("dst_images", "png"), ("src_images", "png")):
            p = os.path.join(out, sub, "%s_%d.%s" % (name, r, ext))
            if not os.path.exists(p) or os.path.getsize(p) == 0:
                return False
    return True


def default_writers(local_world):
    """The rank's share of the node's logical CPUs minus the six its other threads use (submitting thread, four decoders, one spare), clamped to 2..32: every
    rank of a node draws on the same host cores.  A pair costs a writer ~19 ms of host time (NS fill 7, deflate 9.6, Up filter 1.3, .flo 0.7: bench.py's
    `generator_host_share_1of8.writer_stages`), i.e. ~53 pairs/s per thread - round 5's cores / (4 x ranks) gave a rank of an 8-rank node 8 writers on its 32
    CPUs: 94 % busy and 19 % below the GPU's rate."""
    try:
        cores = len(os.sched_getaffinity(0))          # the CPUs this process may run on (a launcher / container may have confined it), not the box's count
    except (AttributeError, OSError):
        cores = os.cpu_count() or 8
    return max(2, min(32, cores // max(1, local_world) - 6))


def self_launch(opt, argv):
    """--gpus N > 1 without a launcher: start one process per GPU under torch.distributed.run and pass the exit status through."""
    import subprocess
    have = torch.cuda.device_count()
    if have < opt.gpus and "MPIFLOW_FORCE_DEVICE" not in os.environ:
        raise SystemExit("gen_3dphoto_dynamic: --gpus %d but only %d GPU(s) are visible on this box" % (opt.gpus, have))
    # --standalone: the launcher's own store picks and keeps a free port (no bind-then-close race); --local-addr: the host name may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(opt.gpus),
           os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    raise SystemExit(subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")).returncode)


def main(argv=None):
    t_main = time.perf_counter()
    opt = parse(argv)
    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(opt, argv)
    if opt.model_engine == "torch" or os.environ.get("MPIFLOW_ENCODER") == "torch":
        # MIOpen in its immediate mode, ONLY for the paths that reach MIOpen (the hip engines run the whole network on this repo's kernels): on a
        # box whose MIOpen user cache is empty the default find mode spends 2.9-3.3 s searching / compiling in the FIRST forward, the immediate
        # mode 0.3 s (tools/miopen_cold_start.py, profiles/r4/generator_startup.txt).  Read by MIOpen at its first convolution; set it yourself
        # to override.  (The fp32 "reference numerics" producer is --model-engine hip --model-dtype fp32, which never touches MIOpen.)
        os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if opt.gpus and opt.gpus != world:
        raise SystemExit("gen_3dphoto_dynamic: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree" % (opt.gpus, world))
    if opt.writers <= 0:
        opt.writers = default_writers(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if rank == 0:
        print(opt)
    # test hooks (multi-process path on a 1-GPU box): MPIFLOW_DIST_BACKEND=gloo, MPIFLOW_FORCE_DEVICE=0
    backend = os.environ.get("MPIFLOW_DIST_BACKEND", "nccl")
    forced = "MPIFLOW_FORCE_DEVICE" in os.environ
    if forced:
        local = int(os.environ["MPIFLOW_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)        # RCCL over xGMI
            if not forced:
                pipeline.assert_distinct_devices(local)                   # one process per GPU, checked before the first collective
        else:
            dist.init_process_group(backend=backend)

    # the host side of a pair is a few batched 3x3 / 4x4 matrix operations: intra-op threading only adds fork / join latency to them (and the
    # writer threads want the cores); per-matrix results do not depend on the thread count
    torch.set_num_threads(1)
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
    model = None
    amp = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(opt.model_dtype) if opt.model_engine == "torch" else None
    if opt.model_engine == "hip" and opt.model_dtype == "bf16":
        raise SystemExit("gen_3dphoto_dynamic: --model-engine hip computes in fp16 (auto), fp32 or fp64; bf16 is a torch autocast dtype (--model-engine torch)")
    if opt.model_engine == "torch" and opt.model_dtype in ("fp64", "fp32-mfma"):
        raise SystemExit("gen_3dphoto_dynamic: --model-dtype %s is the HIP precise engine's (--model-engine hip)" % opt.model_dtype)
    precise_dtype = {"fp32": torch.float32, "fp32-mfma": torch.float32, "fp64": torch.float64}.get(opt.model_dtype) if opt.model_engine == "hip" else None
    if precise_dtype is not None and rank == 0 and opt.mpi_from == "model":
        print("note: --model-engine hip --model-dtype %s selects the PARITY-GRADE producer (mpf_pconv, ~5x the fast engine's time per image, eager launches); "
              "`--model-dtype auto` / fp16 is the fast fp16-storage engine - until round 4 `fp32` named that one (INTEGRATION.md)" % opt.model_dtype, flush=True)
    if opt.mpi_from == "model":                                            # the reference's only producer (:52-60, :92-93)
        from mpiflow_amd.model import MPIPredictor
        if opt.ckpt_path.startswith("random:"):
            model = MPIPredictor(opt.width, opt.height, opt.planes).randomize_(int(opt.ckpt_path.split(":")[1])).eval().to(dev)
        else:
            if not os.path.exists(opt.ckpt_path):
                raise SystemExit("gen_3dphoto_dynamic: checkpoint %r not found.  The reference always runs the AdaMPI network from --ckpt_path "
                                 "(gen_3dphoto_dynamic_v2.py:52-60); pass the checkpoint, or --ckpt_path random:SEED, or choose another "
                                 "producer explicitly with --mpi-from npz|disparity." % opt.ckpt_path)
            model = MPIPredictor.from_checkpoint(opt.ckpt_path, opt.width, opt.height).to(dev)
            opt.planes = model.num_planes
    elif opt.mpi_from == "disparity" and rank == 0:
        print("WARNING: --mpi-from disparity is a stand-in producer (hard depth assignment), not the reference's AdaMPI network", file=sys.stderr)
    use_hip_model = model is not None and opt.model_engine == "hip"
    fill_mode = U.resolve_inpaint(opt.inpaint)
    if fill_mode in ("ns", "telea"):
        fill_mode, fill_algo = "builtin", fill_mode
    else:
        fill_algo = "ns"                                                   # utils/utils.py:284-286
    if fill_mode == "cv2":
        import cv2

        def host_fill(frame, hole):
            return cv2.inpaint(frame, hole, 3, cv2.INPAINT_TELEA if fill_algo == "telea" else cv2.INPAINT_NS)
    elif fill_mode == "builtin":
        def host_fill(frame, hole):
            return ops.inpaint_host(frame, hole, 3, ops.INPAINT_TELEA if fill_algo == "telea" else ops.INPAINT_NS)
    else:
        host_fill = None

    class Lane:
        """Everything one in-flight image owns: its streams, the blended plane stack, the network graph and its static buffers.
        Built ON the lane's stream, so that the zero-fill of the stack and the packed weights are ordered before its first use."""

        def __init__(self):
            self.stream, self.tail_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.stream):
                self.renderer = pipeline.PairRenderer(opt.planes, opt.height, opt.width, dev)
                self.fill_ws = torch.empty(int(_lib.load().mpf_fill_holes_workspace(opt.height, opt.width)), dtype=torch.uint8, device=dev)
                self.inputs = dict(image=torch.empty((3, opt.height, opt.width), device=dev), disp=torch.empty((opt.height, opt.width), device=dev))
                self.hip_model = None
                if use_hip_model and precise_dtype is not None:
                    from mpiflow_amd.model.precise import PrecisePredictor
                    self.hip_model = PrecisePredictor(model, dtype=precise_dtype, x3=opt.model_dtype == "fp32")      # the accuracy mode: fp32 / fp64 on mpf_pconv
                elif use_hip_model:
                    from mpiflow_amd.model.engine import HipPredictor
                    self.hip_model = HipPredictor(model, graph=True)
            self.tail_stream.wait_stream(self.stream)

    lanes = [Lane() for _ in range(max(1, opt.lanes))]
    dstats = pipeline.DeviceStats(dev)
    ring = io_formats.OutputRing(opt.height, opt.width, dev, slots=max(4, 2 * max(opt.writers, 1)), threads=max(opt.writers, 1), host_fill=host_fill)
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

    # mask.max() of every image (the instance-id draws need it): decoded once on rank 0 and broadcast; single rank: read off the
    # masks as they are decoded
    table = pipeline.mask_max_table(names, mask_base, rank, world)
    owned = [i for i in range(len(names)) if (i % world) == rank]
    done_before = set(i for i in owned if opt.resume and outputs_exist(out, names[i].split(".")[0], opt.repeat))
    inputs = iter(io_formats.InputPrefetcher(names, img_base, disp_base, mask_base, [i for i in owned if i not in done_before]))
    skipped, n_resumed = [], 0                       # (image name, reason) of the images this rank owned and could not render
    pending = []                                     # (name, pairs, hand_off) of the image whose pairs are rendered but not yet handed to the writers

    def finish_pending():
        done = 0
        while pending:
            nm, n_new, hand_off = pending.pop(0)
            try:
                hand_off()
            except Exception as e:                                         # noqa: BLE001
                torch.cuda.synchronize()
                skipped.append((nm, "hand-off after %d of %d pairs: %r" % (hand_off.submitted[0], n_new, e)))
            done += hand_off.submitted[0]                                  # what reached the writers (and the statistics) is counted, whatever came after
        return done
    n_pairs, t_first, n_first, n_owned = 0, None, 0, 0
    for i, img in enumerate(names):
        name = img.split(".")[0]
        mine = (i % world) == rank
        item = None
        if mine and i not in done_before:
            with lap("wait for decoded inputs"):
                item = next(inputs)
            assert item["i"] == i
        # ---- the draws of this image, on every rank (RNG schedule contract: an image consumes its 2 x repeat pose draws and its
        #      `repeat` instance-id draws iff its mask decodes and holds at least one instance - the reference dies on such an image,
        #      np.random.randint(0) at :101, before drawing anything.  Nothing else about an image - a broken picture, a render error,
        #      --resume - changes what the other images get.)
        if table is not None:
            mask_max = table[i]
        elif item is not None and item["error"] is None:
            mask_max = int(item["ids_u8"].max())
        else:                                                              # single rank, image resumed or undecodable: the mask alone
            mask_max = io_formats.mask_max_of_file(os.path.join(mask_base, img))
        if mask_max <= 0:
            if mine:
                skipped.append((name, "mask unreadable" if mask_max < 0 else "mask holds no instance (the reference raises here: np.random.randint(0))"))
            continue
        with lap("pose draws"):
            obj_indices, pose_params = [], []
            for r in range(opt.repeat):
                obj_indices.append(np.random.randint(mask_max) + 1)                                                 # :101
                pose_params.append(host_math.draw_pose_parameters(opt.ext_cz, profile=opt.poses))                     # utils.py:207
                pose_params.append(host_math.draw_pose_parameters(opt.ext_cz, base_motions=[0, 0, 0], profile=opt.poses))   # :208
        if not mine:
            continue
        if i in done_before:
            n_resumed += 1
            continue
        if item["error"] is not None:
            skipped.append((name, "input: %r" % (item["error"],)))
            continue
        lane = lanes[n_owned % len(lanes)]
        n_owned += 1
        try:
            n_new, hand_off = render_image(opt, dev, K, out, name, item, obj_indices, pose_params, lane, model, amp, ring, dstats, fill_mode, lap)
        except Exception as e:                                             # noqa: BLE001 - isolate the image, keep the batch going
            torch.cuda.synchronize()
            skipped.append((name, "render: %r" % (e,)))
            n_pairs += finish_pending()                                    # the previous image is unaffected: its pairs still go out
            continue
        n_pairs += finish_pending()                                        # the PREVIOUS image's pairs go to the writers now, behind this image's launches
        pending.append((name, n_new, hand_off))
        if t_first is None and n_owned >= len(lanes):
            t_first, n_first = time.perf_counter(), n_pairs + sum(q[1] for q in pending)       # start-up (graph capture, first launches) ends once every lane has run
            if rank == 0:
                print("start-up: %.2f s from process start until the first image was submitted (imports %.2f s, set-up + model + weight packing "
                      "%.2f s, first image incl. the graph capture %.2f s)" % (
                          t_first - _T_PROCESS, t_main - _T_PROCESS, t_start - t_main, t_first - t_start))
    n_pairs += finish_pending()
    with lap("drain writers"):
        torch.cuda.synchronize()
        ring.close()
    if rank == 0:
        # where the writer threads' time went: the number that says whether 1/N of a node's host cores keeps up with one GPU (bench.py: generator_host_share_1of8)
        rep = ring.stage_report()
        per = ", ".join("%s %.2f ms x %d" % (k, 1e3 * v[0] / max(1, v[1]), v[1]) for k, v in sorted(rep["stages"].items()))
        print("writers: %d threads on %d cpu(s), busy %.0f %% of their wall time, submit() waited %.2f s for a free slot; per call: %s" % (
            rep["threads"], len(os.sched_getaffinity(0)), 100 * rep["busy_share"], rep["backpressure_seconds"], per))
    if lap.on and rank == 0:
        for k, v in prof.items():
            print("  %-40s %8.3f s" % (k, v))
    stats = dstats.result(n_pairs)
    t_end = time.perf_counter()
    stats["wall_seconds"] = t_end - t_start
    if t_first is not None and n_pairs > n_first and rank == 0:
        print("steady state after the first image: %.1f pairs/s on this rank" % ((n_pairs - n_first) / (t_end - t_first)))
    total = pipeline.reduce_stats(stats)
    all_skipped, resumed = pipeline.gather_reports(skipped, n_resumed)
    if rank == 0:
        print("pairs %d  mean|flow| %.3f px  max|flow| %.2f px  hole px/pair %.0f  wall %.1f s  (%d rank%s)" % (
            total["pairs"], total["sum_flow_mag"] / max(total["pairs"], 1) / (opt.height * opt.width), total["max_flow_mag"],
            total["hole_px"] / max(total["pairs"], 1), total["wall_seconds"], world, "s" if world > 1 else ""))
        if resumed:
            print("resume: %d image(s) already complete, skipped" % resumed)
        with open(os.path.join(out, "skipped.txt"), "w") as f:
            for nm, why in all_skipped:
                f.write("%s\t%s\n" % (nm, why))
        if all_skipped:
            print("skipped %d image(s) (listed in %s):" % (len(all_skipped), os.path.join(out, "skipped.txt")))
            for nm, why in all_skipped[:20]:
                print("  %s: %s" % (nm, why))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def render_image(opt, dev, K, out, name, item, obj_indices, pose_params, lane, model, amp, ring, dstats, fill_mode, lap):
    """One owned image: upload, input stage, MPI producer + blend (once), then `repeat` pairs.  Returns (pairs rendered, hand_off): the pairs are
    rendered (enqueued) on return; hand_off() enqueues their way out (statistics, scanlines / hole-fill inputs, device->host copies, writer jobs)."""
    renderer, hip_model, tail_stream, fill_ws = lane.renderer, lane.hip_model, lane.tail_stream, lane.fill_ws
    H, W = opt.height, opt.width
    with torch.cuda.stream(lane.stream):                                  # everything this image enqueues goes to its lane's stream
        with lap("upload + resize image, disparity"):
            rgb8 = item["rgb_u8"].to(dev, non_blocking=True)
            dsp8 = item["disp_u8"].to(dev, non_blocking=True)
            ids = item["ids_u8"].to(dev, non_blocking=True)
            if rgb8.shape[:2] == dsp8.shape[:2]:
                pre = ops.prepare_inputs(rgb_u8=rgb8, disp_u8=dsp8, size=(H, W), out=lane.inputs)     # :82-89 in one launch
            else:                                                          # files of different sizes: each resized on its own, as :86-89 does
                pre = dict(image=ops.prepare_inputs(rgb_u8=rgb8, size=(H, W), out=lane.inputs)["image"],
                           disp=ops.prepare_inputs(disp_u8=dsp8, size=(H, W), out=lane.inputs)["disp"])
            image, disp = pre["image"][None], pre["disp"][None, None]
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
        poses = host_math.poses_from_parameters(pose_params)               # the image's 2 x repeat poses in one batched evaluation
        with lap("instance masks"):
            obj_masks = [ops.prepare_inputs(ids_u8=ids, obj_index=k, size=(H, W))["mask"] for k in obj_indices]      # :102-105
        with lap("render pairs"):
            # utils.py:207-208 draws the dynamic pose first; the camera pose renders with obj_mask, the dynamic one with 1 - obj_mask
            results = renderer.run_pairs(mpi, image[0], K, planes, obj_masks, [(poses[2 * r + 1], poses[2 * r]) for r in range(opt.repeat)],
                                         cum_mask=cum_mask)
        # the tail of the pairs (scanlines / hole-fill hand-off, statistics, copies to the host) runs on a second stream, so it
        # overlaps the next image's network and render - and the HOST enqueues it only after it has submitted the next image's
        # network (main loop): handing five pairs to the writers takes the submitting thread ~1.5 ms, during which the main stream
        # used to run dry at every image boundary (profiles/r3/generator_device_busy.txt: ~3 idle gaps of 0.1 ms per image)
        ready = torch.cuda.Event()
        ready.record()

    submitted = [0]                                                        # pairs whose statistics were added AND whose files were handed to the writers

    def hand_off():
        with torch.cuda.stream(tail_stream):
            tail_stream.wait_event(ready)
            for r, res in enumerate(results):
                for tns in (res["frame_mix"], res["fill_mask"], res["flow_mix"]) + ((res["slab"],) if res.get("slab") is not None else ()):
                    tns.record_stream(tail_stream)
                flo_path, png_path = os.path.join(out, "flows", f"{name}_{r}.flo"), os.path.join(out, "dst_images", f"{name}_{r}.png")   # :120-121
                with lap("hole fill / PNG scanlines + hand-off to the writers"):
                    if fill_mode in ("cv2", "builtin"):                    # :284-286 on the host, on a writer thread
                        ring.submit_pair_fill(res["flow_mix"], res["frame_mix"], res["fill_mask"], flo_path, png_path, slab=res.get("slab"))   # one D2H copy per pair
                    else:
                        frame = ops.fill_holes(res["frame_mix"], res["fill_mask"], workspace=fill_ws) if fill_mode == "peel" else res["frame_mix"]
                        ring.submit_pair(res["flow_mix"], ops.png_scanlines(frame), flo_path, png_path)
                    dstats.add(res["flow_mix"], res["fill_mask"])          # statistics only for pairs that went out: n_pairs and the sums stay consistent
                    submitted[0] += 1
    hand_off.submitted = submitted
    return opt.repeat, hand_off


if __name__ == "__main__":
    main()
