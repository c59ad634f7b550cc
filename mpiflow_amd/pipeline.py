"""Fused per-image pipeline on one GPU + image-sharded multi-GPU driver.

`PairRenderer` is the fast path behind `utils.utils.render_3dphoto_dynamic`:

    Stage A+C  mpf_src_blend_flow   planar [S,4,H,W] stack + image -> interleaved blended RGBA stack + 1-2 flows
                                    (+ fused by-products: source frame as u8 BGR, mask quads of obj_mask / 1 - obj_mask)
    Stage B    mpf_warp_composite   x1 (camera-only) or x2 (object + background poses)  (+ rendered frame as u8 BGR)
    Stage D    mpf_merge            threshold / select / uint8 BGR / fill mask

All buffers are allocated once per (S,H,W) and reused; per pair only ~10 KB of small matrices are uploaded.  With
288 GB of HBM per MI355X a rank can keep hundreds of 629 MB plane stacks resident, so the driver batches images
per rank and never exchanges tensor data between GPUs: images are independent (reference loop
gen_3dphoto_dynamic_v2.py:78-122), the only collective is one all-reduce of a ~10-float statistics vector per batch.
"""
import os
import random

import torch

from . import host_math, ops

MASK_THRESH = 0.99


class _PairHostSide:
    """Host side of a renderer: the small matrices of a pair (K^-1, plane depths, per-plane homographies of every pose) computed with
    the reference's batched torch-CPU expressions and uploaded as d_params blocks.  Needs self.device."""

    # -- host side: small matrices ---------------------------------------------------------------------------------
    def _constants(self, K, disparity):
        """K^-1 and the plane depths are per-image constants: recomputed only when K / the disparities change (keyed by the
        tensors' identity and version, so an in-place edit invalidates them)."""
        key = (id(K), K._version, id(disparity), disparity._version) if isinstance(K, torch.Tensor) and isinstance(disparity, torch.Tensor) else None
        if key is None or getattr(self, "_const_key", None) != key:
            self._const = (host_math.k_inverse(K), host_math.plane_depths(disparity))
            self._const_key = key
            self._const_refs = (K, disparity)          # keep the keyed objects alive so their ids cannot be recycled
        return self._const

    def prepare(self, K, disparity, poses):
        """poses: list of 4x4 G_tgt_src (1 or 2).  Computes K^-1, plane depths, per-plane homographies on the host
        (torch-CPU, reference expressions) and uploads them.  Returns a dict handed to run()."""
        k_inv, d = self._constants(K, disparity)
        H_ts, H_st = host_math.homographies_multi(poses, k_inv, K, d)             # all poses of the pair in one batched evaluation
        wp = [ops.upload_params(ops.warp_params(H_st[i], k_inv, G, d), self.device) for i, G in enumerate(poses)]
        bf, P = ops.blend_flow_params(k_inv, d, H_ts)
        return dict(P=P, blend=ops.upload_params(bf, self.device), warp=wp, k_inv=k_inv, depths=d)

    def prepare_many(self, K, disparity, pose_pairs):
        """prepare() for the R pairs of one image at once: ONE batched homography evaluation over the 2R poses (bit-identical to the
        per-pair ones: every matrix goes through the same per-matrix code) and ONE pinned buffer / H2D copy for the 3R parameter
        blocks (each block starts on a 256-byte boundary), filled by a dozen strided assignments (pack_pair_blocks) instead of 3R
        pack_params calls.  -> R dicts like prepare()'s."""
        k_inv, d = self._constants(K, disparity)
        flat = [G for pair in pose_pairs for G in pair]
        H_ts, H_st = host_math.homographies_multi(flat, k_inv, K, d)
        host, offs, sizes = pack_pair_blocks(k_inv, d, H_ts, H_st, flat, pin=True)
        dev = host.to(device=self.device, non_blocking=True)
        view = lambda r, j: dev[offs[r][j]:offs[r][j] + sizes[j]]  # noqa: E731
        return [dict(P=2, blend=view(r, 0), warp=[view(r, 1), view(r, 2)], k_inv=k_inv, depths=d) for r in range(len(pose_pairs))]


def pack_pair_blocks(k_inv, depths, H_ts, H_st, poses, pin=False):
    """Host image of the parameter blocks of R pairs in one buffer: per pair [blend_flow_params(k_inv, d, H_ts[2r:2r+2]) |
    warp_params(H_st[2r], k_inv, poses[2r], d) | warp_params(H_st[2r+1], k_inv, poses[2r+1], d)], every block padded to a multiple of
    64 floats.  H_ts / H_st: [2R,S,3,3], poses: 2R [4,4].  The same floats ops.blend_flow_params / ops.warp_params pack one block at a
    time (tests/test_host_logic.py compares them).  -> (buffer [R * stride] f32, offsets[r] = (blend, warp0, warp1), sizes (blend, warp, warp))"""
    HDR, REC = host_math.PARAMS_HEADER, host_math.PLANE_RECORD
    d = host_math._cpu32(depths).reshape(-1)
    S, R = d.numel(), len(poses) // 2
    nb, nw = HDR + REC * 2 * S, HDR + REC * S
    ab, aw = (nb + 63) // 64 * 64, (nw + 63) // 64 * 64
    stride = ab + 2 * aw
    host = torch.zeros(R * stride, dtype=torch.float32)
    if pin:
        host = host.pin_memory()
    rows = host.view(R, stride)
    k9 = host_math._cpu32(k_inv).reshape(9)
    G = torch.stack([host_math._cpu32(g).reshape(4, 4)[0:3, :].reshape(12) for g in poses]).view(R, 2, 12)
    Hts = host_math._cpu32(H_ts).reshape(R, 2, S, 9)
    Hst = host_math._cpu32(H_st).reshape(R, 2, S, 9)
    rows[:, 0:9] = k9
    rec = rows[:, HDR:HDR + REC * 2 * S].view(R, S, 2, REC)                 # record = s * P + p
    rec[..., 0:9] = Hts.permute(0, 2, 1, 3)
    rec[..., 9] = d.view(1, S, 1)
    for v in range(2):
        o = ab + v * aw
        rows[:, o:o + 9] = k9
        rows[:, o + 9:o + 21] = G[:, v]
        rec = rows[:, o + HDR:o + HDR + REC * S].view(R, S, REC)
        rec[..., 0:9] = Hst[:, v]
        rec[..., 9] = d.view(1, S)
    return host, [(r * stride, r * stride + ab, r * stride + ab + aw) for r in range(R)], (nb, nw, nw)


class PairRenderer(_PairHostSide):
    """Preallocated fused renderer for one (S, H, W) on one device."""

    def __init__(self, S, H, W, device, n_views=2, with_depth=False):
        self.S, self.H, self.W, self.device = S, H, W, torch.device(device)
        f32 = torch.float32
        dev = self.device
        self.rgba = ops.alloc_rgba_stack(S, H, W, dev)                            # blended interleaved stack (+ tail padding)
        self.flows = torch.empty((n_views, 2, H, W), dtype=f32, device=dev)
        # the fused pipeline never reads depth / tgt_mask (the reference discards them too, utils/utils.py:210, :330):
        # leaving them out selects the leaner Stage B body
        self.views = [dict(rgb=torch.empty((3, H, W), dtype=f32, device=dev), objmask=torch.empty((H, W), dtype=f32, device=dev),
                           rgb_u8=torch.empty((H, W, 3), dtype=torch.uint8, device=dev)) for _ in range(n_views)]
        if with_depth:                       # the depth-ordered variant ("utils/utils copy.py":295-303) reads both views' composited depth
            for v in self.views:
                v["depth"] = torch.empty((H, W), dtype=f32, device=dev)
        self.quads = [torch.empty((H, W, 4), dtype=f32, device=dev) for _ in range(2)]    # obj_mask, 1 - obj_mask
        self.src_u8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        self.n_views = n_views
        # all Stage B views of a pair in ONE launch (mpf_warp_composite_views): with the strip-major tile order it wins at every shape
        # measured - x1.06 (128 x 1024 x 1536) to x1.17 (64 x 640 x 960) for a pair, x1.20-1.42 for the 10 views of a `repeat` loop
        # (profiles/r2/stage_b_views_strip_shapes.log).  False = one launch per view (bench.py's comparison record, tests).
        self.multi_view = True
        self._pair_bufs = []

    # -- device side: launches only ------------------------------------------------------------------------------------
    def blend(self, mpi, image, K, disparity, cum_mask=None):
        """Blend the source image into the stack once per IMAGE (the blended stack does not depend on the pose): the
        reference re-blends inside every render_3dphoto_dynamic call, i.e. `repeat` times per image
        (gen_3dphoto_dynamic_v2.py:99-118); with this, every further pair of the same image runs Stage A+C flow-only
        (reads 4*S*N instead of 16*S*N, writes nothing but the flows)."""
        k_inv, d = self._constants(K, disparity)
        ops.src_blend_flow(mpi, image, K_inv=k_inv, depth_S=d, homs_tgt_src=None, out_rgba=self.rgba, src_u8=self.src_u8, cum_mask=cum_mask)

    def run(self, mpi, image, prep, obj_mask, complement=(False, True), cum_mask=None, reuse_blend=False):
        """mpi [S,4,H,W], image [3,H,W], obj_mask [H,W] on device.  Two or three launches:
          Stage A+C (+ source frame as u8, + mask quads of obj_mask and 1 - obj_mask), then one Stage B per view
          (view v samples 1 - obj_mask when complement[v]; + its frame as u8).  cum_mask [S,H,W]: `mpi` is the raw decoder
          output of the AdaMPI network and its activation epilogue is fused into Stage A+C.  reuse_blend: self.rgba / self.src_u8
          already hold this image's blended stack (blend()), Stage A+C only computes flows and quads.
          Returns (flows [P,2,H,W], views)."""
        P = prep["P"]
        need_c = any(complement[:P])
        need_p = not all(complement[:P])
        ops.src_blend_flow(mpi, image, out_rgba=None if reuse_blend else self.rgba, want_rgba=False, out_flows=self.flows[:P],
                           dparams=prep["blend"], P=P, src_u8=None if reuse_blend else self.src_u8, obj_mask=obj_mask,
                           quads=self.quads[0] if need_p else None, quads_complement=self.quads[1] if need_c else None,
                           cum_mask=cum_mask)
        if P > 1 and self.multi_view:
            ops.warp_composite_views(self.rgba, [dict(dparams=prep["warp"][v], quads=self.quads[1 if complement[v] else 0], out=self.views[v])
                                                 for v in range(P)], interleaved=2)
        else:
            for v in range(P):
                ops.warp_composite(self.rgba, self.quads[1 if complement[v] else 0], dparams=prep["warp"][v], out=self.views[v],
                                   interleaved=2)
        return self.flows, self.views


    # -- all pairs of one image --------------------------------------------------------------------------------------------
    def _pair_buffers(self, n):
        f32, dev, H, W = torch.float32, self.device, self.H, self.W
        while len(self._pair_bufs) < n:
            self._pair_bufs.append(dict(
                flows=torch.empty((2, 2, H, W), dtype=f32, device=dev), quads=[torch.empty((H, W, 4), dtype=f32, device=dev) for _ in range(2)],
                views=[dict(rgb=torch.empty((3, H, W), dtype=f32, device=dev), objmask=torch.empty((H, W), dtype=f32, device=dev)) for _ in range(2)]))
        return self._pair_bufs[:n]

    def run_pairs(self, mpi, image, K, disparity, obj_masks, poses, cum_mask=None, thresh=MASK_THRESH):
        """The `repeat` pairs of ONE image (gen_3dphoto_dynamic_v2.py:99-118) whose blended stack is already in self.rgba (blend()):
        per pair a flow-only Stage A+C (both flows + the mask quads), then ALL their posed views - 2 per pair - in as few Stage B
        launches as 16 views per launch allow, then a merge per pair.  obj_masks: R tensors [H,W]; poses: R (G_cam, G_dyn) tuples.
        Same results, bit for bit, as R calls of render_pair(..., reuse_blend=True).  Returns R dicts(flow_mix, frame_mix, fill_mask, slab):
        the three products are views of `slab` (ops.pair_slab)."""
        R = len(obj_masks)
        bufs = self._pair_buffers(R)
        views = []
        preps = self.prepare_many(K, disparity, [[G_cam, G_dyn] for (G_cam, G_dyn) in poses])
        for om, prep, b in zip(obj_masks, preps, bufs):
            ops.src_blend_flow(mpi, image, out_rgba=None, want_rgba=False, out_flows=b["flows"], dparams=prep["blend"], P=2, obj_mask=om,
                               quads=b["quads"][0], quads_complement=b["quads"][1], cum_mask=cum_mask)
            views += [dict(dparams=prep["warp"][v], quads=b["quads"][v], out=b["views"][v]) for v in range(2)]
        if self.multi_view:
            for i in range(0, len(views), 16):
                ops.warp_composite_views(self.rgba, views[i:i + 16], interleaved=2)
        else:
            for v in views:
                ops.warp_composite(self.rgba, v["quads"], dparams=v["dparams"], out=v["out"], interleaved=2)
        out = []
        for om, b in zip(obj_masks, bufs):
            v = b["views"]
            slab, views3 = ops.pair_slab(self.H, self.W, self.device)      # the three products in one buffer: they leave the GPU in one copy
            flow_mix, frame_mix, fill = ops.merge(v[0]["rgb"], v[1]["rgb"], v[0]["objmask"], v[1]["objmask"], b["flows"][0], b["flows"][1], om, thresh, out=views3)
            out.append(dict(flow_mix=flow_mix, frame_mix=frame_mix, fill_mask=fill, slab=slab))
        return out


class OverlappedPairRenderer(_PairHostSide):
    """A STREAM of dynamic pairs (one image each) rendered as a two-stage software pipeline: Stage B (both posed views) of pair i and
    Stage A+C (blend + both flows + source frame + mask quads) of pair i+1 go into ONE launch whose grid interleaves the two kinds of
    workgroups (mpf_warp_views_and_blend_next), so the HBM-bound source-frame pass runs underneath the issue-bound target-frame
    passes instead of in front of them.  Per pair: one fused launch + one merge; the first pair of a run pays a stand-alone Stage A+C,
    the last one (flush()) a stand-alone Stage B.  Same kernels bodies, same results, bit for bit, as PairRenderer.run() + merge.

    push(mpi, image, prep, obj_mask, out) enqueues pair i+1 and completes pair i (its `out` = (flow_mix [H,W,2], frame_mix [H,W,3] u8,
    fill_mask [H,W] u8) is written, stream-ordered, by the time push returns); flush() completes the last one.  Two slots of
    per-pair buffers (blended stack, flows, quads, views) alternate: a slot is rewritten only after its pair has been merged.

    LIFETIME of the caller's tensors: `mpi`, `image` and `obj_mask` are consumed by the push() they are given to (stream-ordered: they may be
    rewritten on the same stream right after it returns) - the deferred merge reads the object mask from the slot's own mask quads, not from
    the caller's tensor.  EXCEPTION - an independent chain (attach_chain(ordered=False)): the chain reads `image` and the `moving` tensors on a SIDE
    stream, ordered only behind `moving_ready`; they must stay untouched until the handed-back set's `.ready` event (or flush()), not just until push() returns.  `out` of pair i (and `moving`'s disparity / instance mask) is written / read by the launches that COMPLETE pair i,
    i.e. inside the NEXT push() / flush() (two further push() calls with merge_in_launch): it must stay untouched until that call has been issued.

    attach_chain(chain): SURVEY 8(d)'s full c3 - the moving-object chain of every pair (moving_obj.MovingObjectChain: depth -> flow
    projection, forward warp of the pair's uint8 source frame, masks; moving_obj.py:29-150) runs on a SIDE stream.  The chain of pair i
    needs nothing but the source frame Stage A+C of pair i wrote, so it is issued right behind the launch that carried that role and runs
    underneath the NEXT launch (Stage B of pair i); the main stream waits for it only when pair i is handed back, a whole pair launch
    later - no wait ever sits between two pair launches.

    merge_in_launch=True: Stage D of a pair rides in a LATER pair launch instead of being a launch of its own between two of them (it is
    8 us of kernel plus two launch boundaries on the critical path): the Stage A+C role of launch i+2 merges pair i as a per-pixel prologue
    (mpf_warp_views_blend_next_merge_prev) - the thread that merges a pixel is the one that later overwrites that pixel's flows in the
    slot the two pairs share, so no further buffering is needed.  The stream is then ONE launch per pair and nothing else; push() hands back
    the pair enqueued TWO calls earlier, `out` of a pair must stay untouched for two further push() calls (three alternating buffers), and
    flush() returns the last two pairs."""

    def __init__(self, S, H, W, device, thresh=MASK_THRESH, merge_in_launch=False):
        self.S, self.H, self.W, self.device, self.thresh = S, H, W, torch.device(device), thresh
        self.merge_in_launch, self._merging, self.chain_ordered = merge_in_launch, None, True
        self.guard_unconsumed = True                                 # see _start_chain_unordered
        f32, dev = torch.float32, self.device
        self.slots = [dict(rgba=ops.alloc_rgba_stack(S, H, W, dev), flows=torch.empty((2, 2, H, W), dtype=f32, device=dev),
                           quads=[torch.empty((H, W, 4), dtype=f32, device=dev) for _ in range(2)],
                           src_u8=torch.empty((H, W, 3), dtype=torch.uint8, device=dev),
                           views=[dict(rgb=torch.empty((3, H, W), dtype=f32, device=dev), objmask=torch.empty((H, W), dtype=f32, device=dev),
                                       rgb_u8=torch.empty((H, W, 3), dtype=torch.uint8, device=dev)) for _ in range(2)])
                      for _ in range(2)]
        self._next, self._pending = 0, None
        self.chain, self.side = None, None
        self.on_fused = None                                         # optional hook(callable launching) -> used by bench.py to bracket with events
        # the overlapped launch addresses the stack through 32-bit buffer offsets: stacks of 4 GiB and more take the two separate launches
        self.fusable = S * H * W * 16 < (1 << 32)

    def attach_chain(self, chain, high_priority=False, ordered=True, cu_stride=0, sides=1):
        """chain: moving_obj.MovingObjectChain with (at least) two output sets - three with merge_in_launch.
        ordered=True: the chain splats the uint8 source frame the pair's Stage A+C role wrote and its results are stream-ordered on the
          MAIN stream when the pair is handed back - costs the main stream an event record and an event wait per pair (measured: 12 us per
          pair at 64 x 640 x 960, the command processor handles both between two pair launches).
        ordered=False: the chain is an independent side pipeline.  It converts the pair's float image to the same uint8 bytes itself, so it
          needs no output of the render path and NOTHING is inserted into the main stream: it is issued at the top of push(), each handed-back
          ops.MovingObjectBuffers carries `.ready` (a torch event to wait for on whatever stream consumes it; `image` / `moving` tensors of
          that pair must stay untouched until then).  A set is rewritten len(chain.bufs) push() calls after it was handed out: a consumer
          on another stream sets `.consumed` (an event) on it and the chain waits for that; without one the chain waits for an event recorded
          on the caller's stream at the push() that reuses the set (`guard_unconsumed`, default on), which covers every consumer on that
          stream.  `moving_ready` of push() names the event after which the pair's inputs may be read (None: one is recorded on the main
          stream), and flush() joins the side stream.
        high_priority: the side stream gets the device's highest stream priority (no measurable effect at 64 x 640 x 960).
        sides (independent chain only): number of side streams the chains alternate over - with 2, the chain of pair i may take TWO pair launches before it
          delays anything (the chain needs `sides` more output sets than otherwise)."""
        # a pair's output set must survive until the pair has been handed back: one push() later, two with merge_in_launch - the sets are used
        # round-robin, so that takes two resp. three of them
        need = (3 if self.merge_in_launch else 2) + (max(1, int(sides)) - 1 if not ordered else 0)
        if len(chain.bufs) < need or (chain.H, chain.W) != (self.H, self.W):
            raise ValueError("attach_chain: the chain needs %d output sets (n_buffers) of %d x %d for this renderer" % (need, self.H, self.W))
        if cu_stride and cu_stride > 1 and high_priority:
            raise ValueError("attach_chain: a CU-masked side stream (cu_stride) has no priority; give one of cu_stride / high_priority")
        if cu_stride and cu_stride > 1 and not ordered and int(sides) > 1:
            raise ValueError("attach_chain: cu_stride masks ONE side stream; with sides > 1 the others would run on every CU - give sides=1 with cu_stride")
        self.close()                                                  # a previously attached chain's CU-masked stream (detaches that chain)
        self.chain, self.chain_ordered, self._chain_next = chain, ordered, 0
        if cu_stride and cu_stride > 1:
            # the side stream may only use every cu_stride-th compute unit: the chain's latency-sized workgroups then sit on few CUs instead of
            # taking a slot here and there on all of them, underneath a launch that fills the whole chip
            import ctypes
            from . import _lib
            h = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().mpf_stream_create_cu_subset(int(cu_stride), 0, ctypes.byref(h)), "mpf_stream_create_cu_subset")
            self._side_handle = h                                     # owned for the renderer's lifetime
            self.side = torch.cuda.ExternalStream(h.value, device=self.device)
        else:
            self.side = torch.cuda.Stream(self.device, priority=-1 if high_priority else 0)
        self._sides = [self.side] + [torch.cuda.Stream(self.device, priority=-1 if high_priority else 0) for _ in range(max(1, int(sides)) - 1 if not ordered else 0)]
        self._chain_count, self._guard_ev = 0, None
        for k, s in enumerate(self.slots):
            s["index"], s["ev_src"], s["ev_chain"], s["moving"] = k, torch.cuda.Event(), torch.cuda.Event(), None

    def close(self):
        """Give back the CU-masked side stream of attach_chain(cu_stride=...) (mpf_stream_create_cu_subset), after it has drained."""
        h = getattr(self, "_side_handle", None)
        if h is not None:
            from . import _lib
            self._side_handle = None
            try:
                self.side.synchronize()
            finally:
                # the wrapper in _sides points at the destroyed stream: detach everything that could launch on it
                self.side, self._sides, self.chain = None, [], None
                _lib.load().mpf_stream_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:                                             # noqa: BLE001 - interpreter shutdown
            pass

    def _views(self, slot, prep):
        return [dict(dparams=prep["warp"][v], quads=slot["quads"][v], out=slot["views"][v]) for v in range(2)]

    def _start_chain(self, slot, moving):
        """ordered chain, right behind the launch whose Stage A+C role wrote slot['src_u8']: the pair's moving-object chain, on the side
        stream.  -> the output set, or None"""
        if self.chain is None or not self.chain_ordered or moving is None:
            return None
        slot["ev_src"].record()
        which, self._chain_next = self._chain_next, (self._chain_next + 1) % len(self.chain.bufs)      # output sets round-robin (see attach_chain)
        with torch.cuda.stream(self.side):
            self.side.wait_event(slot["ev_src"])
            if os.environ.get("MPF_CHAIN_DEBUG") == "events_only":      # measurement hook (tools/): the event choreography without the chain's kernels
                b = self.chain.bufs[which]
            else:
                b = self.chain.run(moving[0], moving[1], slot["src_u8"], which=which)
            slot["ev_chain"].record()
        slot["chain_busy"] = True
        return b

    def _start_chain_unordered(self, image, moving, moving_ready):
        """independent chain, at the top of push(): reads nothing the render path writes.  -> the output set, or None"""
        if self.chain is None or self.chain_ordered or moving is None:
            return None
        which, self._chain_next = self._chain_next, (self._chain_next + 1) % len(self.chain.bufs)
        b = self.chain.bufs[which]
        if moving_ready is None:
            moving_ready = torch.cuda.Event()
            moving_ready.record()
        guard = None
        if self.guard_unconsumed:
            # A set that was handed out before and whose consumer left no `consumed` event: everything enqueued on the CALLER's stream until shortly
            # before this push() finishes before the chain rewrites the set - a consumer on that stream is safe without doing anything, one on another
            # stream must set `consumed`.  A set used at push u is handed back at push u + d (d = 2 with merge_in_launch, else 1) and rewritten at push
            # u + n (n sets): any event recorded on the caller's stream at a push in [u + d + 1, u + n] does, so ONE event every n - d pushes serves
            # every set (an event record on the pair stream costs 0.5 % of the pair rate when recorded at every push).
            w = max(1, len(self.chain.bufs) - (2 if self.merge_in_launch else 1))
            if self._chain_count % w == 0 or self._guard_ev is None:
                self._guard_ev = torch.cuda.Event()
                self._guard_ev.record()
            if b.consumed is None and getattr(b, "ready", None) is not None:
                guard = self._guard_ev
        side = self._sides[self._chain_count % len(self._sides)]
        self._chain_count += 1
        with torch.cuda.stream(side):
            side.wait_event(moving_ready)
            if len(self._sides) > 1 and getattr(b, "ready", None) is not None:
                side.wait_event(b.ready)                                   # the set's previous chain may have run on another side stream
            if b.consumed is not None:
                side.wait_event(b.consumed)
                b.consumed = None
            elif guard is not None:
                side.wait_event(guard)
            self.chain.run(moving[0], moving[1], image, which=which)
            b.ready = torch.cuda.Event()
            b.ready.record()
        return b

    def _wait_chain_of(self, slot):
        """main stream: the ordered chain that reads this slot's source frame has finished (issued at least a whole pair launch ago)."""
        if slot.get("chain_busy"):
            torch.cuda.current_stream().wait_event(slot["ev_chain"])
            slot["chain_busy"] = False

    def _out_of(self, pend):
        if pend["out"] is None:
            pend["out"] = ops.pair_slab(self.H, self.W, self.device)[1]           # one buffer: the three products can leave in one copy
        return pend["out"]

    def _merge_operands(self, pend):
        """The merge of a pair reads ONLY buffers of the pair's slot: its object mask is the .x of the mask quads its Stage A+C role wrote
        (quads[n].x == obj_mask[n], stride 4), so the caller's mask tensor is consumed by the push() it is given to."""
        slot = pend["slot"]
        v = slot["views"]
        return (v[0]["rgb"], v[1]["rgb"], v[0]["objmask"], v[1]["objmask"], slot["flows"][0], slot["flows"][1], slot["quads"][0])

    def _handed_back(self, pend):
        done = tuple(pend["out"])
        if pend["moving"] is not None:
            if self.chain_ordered:
                self._wait_chain_of(pend["slot"])
            done = done + (pend["moving"],)
        return done

    def _finish(self, pend):
        """Stage D as a launch of its own, then the pair is handed back."""
        ops.merge(*self._merge_operands(pend), self.thresh, out=self._out_of(pend), obj_mask_stride=4)
        return self._handed_back(pend)

    def _blend_alone(self, slot, mpi, image, prep, obj_mask, cum_mask):
        ops.src_blend_flow(mpi, image, out_rgba=slot["rgba"], out_flows=slot["flows"], dparams=prep["blend"], P=2, src_u8=slot["src_u8"],
                           obj_mask=obj_mask, quads=slot["quads"][0], quads_complement=slot["quads"][1], cum_mask=cum_mask)

    def push(self, mpi, image, prep, obj_mask, out=None, cum_mask=None, moving=None, moving_ready=None):
        """Enqueue one pair (prep from prepare(K, disparity, [G_cam, G_dyn]); view 0 samples obj_mask, view 1 its complement).
        moving = (disp [H,W], instance mask [H,W]) with a chain attached: the pair's moving-object chain runs on the side stream.
        Returns the (flow_mix, frame_mix, fill_mask[, ops.MovingObjectBuffers]) of the pair this call completed - the PREVIOUS one, or with
        merge_in_launch the one before that - or None if there was none."""
        assert prep["P"] == 2
        slot = self.slots[self._next]
        self._next ^= 1
        done = None
        obj_mask = obj_mask.reshape(self.H, self.W)
        if obj_mask.dtype != torch.float32 or not obj_mask.is_contiguous():
            obj_mask = obj_mask.to(torch.float32).contiguous()
        new = dict(slot=slot, prep=prep, out=out, moving=self._start_chain_unordered(image, moving, moving_ready))
        self._wait_chain_of(slot)                                    # this slot's source frame is about to be rewritten
        pend = self._pending
        if pend is None:
            self._blend_alone(slot, mpi, image, prep, obj_mask, cum_mask)
        elif not self.fusable:
            ops.warp_composite_views(pend["slot"]["rgba"], self._views(pend["slot"], pend["prep"]), interleaved=2)
            self._blend_alone(slot, mpi, image, prep, obj_mask, cum_mask)
        else:
            mg, self._merging = self._merging, None
            mp = ops.merge_args(*self._merge_operands(mg), self.thresh, self._out_of(mg), obj_mask_stride=4) if mg is not None else None
            launch = lambda: ops.warp_views_and_blend_next(                                                   # noqa: E731
                pend["slot"]["rgba"], self._views(pend["slot"], pend["prep"]), mpi, image, prep["blend"], 2, slot["rgba"],
                out_flows_next=slot["flows"], src_u8_next=slot["src_u8"], obj_mask_next=obj_mask, quads_next=slot["quads"][0],
                quads_complement_next=slot["quads"][1], cum_mask_next=cum_mask, merge_prev=mp)
            if self.on_fused is not None:
                self.on_fused(launch)
            else:
                launch()
            if mg is not None:
                done = self._handed_back(mg)
        if new["moving"] is None:
            new["moving"] = self._start_chain(slot, moving)
        if pend is not None:
            if self.merge_in_launch and self.fusable:
                self._merging = pend                                 # its Stage B is in flight; its merge rides in the next launch
            else:
                done = self._finish(pend)
        self._pending = new
        return done

    def flush(self):
        """Complete what is in flight: the last pair's stand-alone Stage B launch and the outstanding merges.  Returns the LIST of the pairs
        completed here, oldest first - each (flow_mix, frame_mix, fill_mask[, chain buffers]): at most one, with merge_in_launch at most two,
        [] when nothing was pending."""
        pend, mg = self._pending, self._merging
        self._pending = self._merging = None
        res = []
        if pend is not None:
            ops.warp_composite_views(pend["slot"]["rgba"], self._views(pend["slot"], pend["prep"]), interleaved=2)
        for p in (mg, pend):
            if p is not None:
                res.append(self._finish(p))
        if self.chain is not None and not self.chain_ordered:
            for side in self._sides:
                torch.cuda.current_stream().wait_stream(side)                 # the independent chain joins the main stream here
        return res

    @property
    def pending_slot(self):
        return None if self._pending is None else self._pending["slot"]


def hard_flows(mpi_S4HW, disparity_S, K, poses, generic=False):
    """hard_flow=True (reference utils/mpi/mpi_rendering.py:126-130): the flow of the arg-max-weight plane instead of the weighted sum, for every pose,
    clipped to +-200 (utils/utils.py:348).  One pass over the stack's sigma planes for all poses (mpf_src_flow_hard, two poses per launch);
    generic=True: the materialised form behind the reference's own functions (per-plane flows written and re-read) - same bits, kept as the witness."""
    S, _, H, W = mpi_S4HW.shape
    dev = mpi_S4HW.device
    k_inv = host_math.k_inverse(K)
    d = host_math.plane_depths(disparity_S)
    if not generic:
        homs = [host_math.homographies(G, k_inv, K, d)[0] for G in poses]
        out = [ops.src_flow_hard(mpi_S4HW, k_inv, d, torch.stack(homs[i:i + 2]), flow_clip=200.0) for i in range(0, len(homs), 2)]
        return torch.cat(out)
    xyz = ops.src_xyz(k_inv, d, H, W, dev)
    out = []
    for G in poses:
        H_ts, _ = host_math.homographies(G, k_inv, K, d)
        per_plane = ops.homography_flow(H_ts, H, W, dev).permute(0, 3, 1, 2).contiguous()
        r = ops.volume_render(None, mpi_S4HW[:, 3].contiguous(), xyz, extra_SEN=per_plane, hard=True, want_tacc=False, want_weights=False)
        out.append(torch.clip(r["extra"], -200, 200))
    return torch.stack(out)


def render_pair(image_3HW, obj_mask_HW, mpi_S4HW, disparity_S, K, G_cam, G_dyn, thresh=MASK_THRESH, renderer=None, cum_mask=None,
                hard_flow=False, reuse_blend=False, depth_ordered=False):
    """Everything render_3dphoto_dynamic does up to the inputs of cv2.inpaint (reference utils/utils.py:159-283), for
    explicit poses: G_cam renders with obj_mask, G_dyn with 1 - obj_mask (sic - SURVEY §3.2).  Device tensors in/out.
    depth_ordered: also the older module's depth-ordered frame ("utils/utils copy.py":295-303) as `frame_mix_depth` (+ `depth_mask`);
    the renderer then needs the views' depth (PairRenderer(with_depth=True))."""
    mpi = mpi_S4HW
    S, _, H, W = mpi.shape
    r = renderer or PairRenderer(S, H, W, mpi.device, with_depth=depth_ordered)
    om = obj_mask_HW.reshape(H, W).to(torch.float32)
    prep = r.prepare(K, disparity_S, [G_cam, G_dyn])
    flows, views = r.run(mpi, image_3HW.reshape(3, H, W), prep, om, cum_mask=cum_mask, reuse_blend=reuse_blend)
    if hard_flow:
        assert cum_mask is None, "hard_flow needs the activated stack"
        flows = hard_flows(mpi, disparity_S, K, [G_cam, G_dyn])
    flow_mix, frame_mix, fill = ops.merge(views[0]["rgb"], views[1]["rgb"], views[0]["objmask"], views[1]["objmask"],
                                          flows[0], flows[1], om, thresh)
    out = dict(flow_mix=flow_mix, frame_mix=frame_mix, fill_mask=fill, src_np=r.src_u8,
               view_cam=views[0], view_dyn=views[1], flows=flows, rgba=r.rgba)
    if depth_ordered:
        if views[0].get("depth") is None or views[1].get("depth") is None:
            raise ValueError("depth_ordered needs a PairRenderer(with_depth=True)")
        out["frame_mix_depth"], out["depth_mask"] = ops.merge_depth_ordered(
            views[0]["rgb"], views[1]["rgb"], views[0]["objmask"], views[1]["objmask"], views[0]["depth"], views[1]["depth"], thresh,
            want_depth_mask=True)
    return out


# ---- multi-GPU: image sharding + end-of-batch statistics ------------------------------------------------------------

def shard_indices(n_items, rank, world_size):
    """Images i with i % world_size == rank; every image is owned by exactly one rank."""
    return list(range(rank, n_items, world_size))


def pose_schedule(seed, n_pairs, ext_cz):
    """Replay of the reference's global RNG stream (gen_3dphoto_dynamic_v2.py:38-39 seeds once; every pair then draws a
    dynamic pose and a camera pose, utils/utils.py:207-208).  Every rank replays the whole schedule and indexes it by
    global pair id, so an N-GPU run produces the same poses as a 1-GPU run."""
    rng = random.Random(seed)
    out = []
    for _ in range(n_pairs):
        dyn = host_math.generate_random_pose(ext_cz, rng=rng)
        cam = host_math.generate_random_pose(ext_cz, base_motions=(0, 0, 0), rng=rng)
        out.append((cam, dyn))
    return out


def mask_max_table(names, mask_dir, rank, world, group=None, threads=8):
    """The instance-id draw of every pair needs mask.max() of ITS image (gen_3dphoto_dynamic_v2.py:101), and every rank replays
    the draws of every image to stay on the reference's RNG stream.  Instead of every rank decoding every mask (N ranks x the
    whole masks/ directory), rank 0 decodes each mask once, and the table of maxima - one int32 per image, -1 = unreadable -
    is broadcast (RCCL under nccl, gloo on CPU).  Ranks then decode only the masks of the images they own."""
    import concurrent.futures
    import torch.distributed as dist
    from . import io_formats
    n = len(names)
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return None                                   # single rank: the maxima come from the decoded masks themselves
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    if rank == 0:
        with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
            vals = list(pool.map(lambda nm: io_formats.mask_max_of_file(os.path.join(mask_dir, nm)), names))
        table = torch.tensor(vals, dtype=torch.int32, device=dev)
    else:
        table = torch.empty((n,), dtype=torch.int32, device=dev)
    dist.broadcast(table, src=0, group=group)
    return table.cpu().tolist()


def gather_reports(skipped, n_resumed, group=None):
    """End of batch: every rank's list of (image, reason) it could not render, and its count of resumed images, on rank 0."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return list(skipped), n_resumed
    world = dist.get_world_size(group)
    box = [None] * world
    dist.all_gather_object(box, (list(skipped), int(n_resumed)), group=group)
    merged = sorted(x for lst, _ in box for x in lst)
    return merged, sum(n for _, n in box)


def device_identity(local_index):
    """A string that is equal for two processes of this job iff they drive the same GPU: host name, the visible-device lists the
    process was started with, and the device index it selected.  (Deliberately not the runtime's uuid field, which some ROCm
    builds leave empty or identical: a false alarm here would abort a healthy multi-GPU run.)"""
    import socket
    env = "|".join(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    return "%s|%s|%d" % (socket.gethostname(), env, int(local_index))


def device_description(local_index):
    """Human-readable record of the GPU a rank drives (bench.py / the CLI put one per rank into their reports): name, PCI address,
    CU count.  Information only - the one-process-per-GPU check keys on device_identity()."""
    try:
        p = torch.cuda.get_device_properties(int(local_index))
        pci = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
        return "cuda:%d %s pci %s %d CUs" % (int(local_index), p.name, pci, p.multi_processor_count)
    except Exception as e:                                   # no GPU (CPU CI): still a record
        return "cuda:%d (%s)" % (int(local_index), type(e).__name__)


_side_groups = {}


def host_side_group(group=None):
    """A gloo (CPU, TCP) process group over the SAME ranks as `group` (default: the world), for the small host-side exchanges that must not
    touch RCCL: under the "nccl" backend a collective between two ranks that share one GPU stalls or aborts deep inside RCCL, which is
    exactly the misconfiguration assert_distinct_devices() exists to report.  Public API only (dist.new_group); created once per group,
    collectively (new_group is a collective over the WORLD: every rank of the job must reach this call, in the same order, for a given
    `group` - also the ranks that are not members, which get the non-member handle back)."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        return group
    world = dist.group.WORLD                                  # a side group belongs to ONE default group: never reuse it after a re-init
    key = (id(world), None if group is None or group is world else id(group))
    hit = _side_groups.get(key)
    if hit is None or hit[0] is not world:
        ranks = None if key[1] is None else dist.get_process_group_ranks(group)
        hit = (world, dist.new_group(ranks=ranks, backend="gloo"), group)      # keep `group` alive so that its id cannot be recycled
        _side_groups[key] = hit
    return hit[1]


def exchange_device_records(local_index, group=None):
    """-> ([identity of rank 0..world-1], [description of rank 0..world-1]) on every rank; one all_gather_object over gloo."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    box = [None] * world
    dist.all_gather_object(box, (device_identity(local_index), device_description(local_index)), group=host_side_group(group))
    return [b[0] for b in box], [b[1] for b in box]


def assert_distinct_devices(local_index, group=None):
    """One process per GPU: every rank publishes the identity of its device over the gloo side group (no RCCL
    communicator is needed, so it also works before - and instead of - the first RCCL collective) and fails fast when two ranks
    share one; RCCL would otherwise stall or abort deep inside its first all-reduce.  Returns the identities."""
    ids, _ = exchange_device_records(local_index, group)
    dup = sorted({i for i in ids if ids.count(i) > 1})
    if dup:
        raise RuntimeError("ranks share a GPU: %s (rank -> device: %s); launch one process per GPU" % (dup, dict(enumerate(ids))))
    return ids


STAT_NAMES = ["pairs", "sum_flow_mag", "hole_px", "kernel_seconds", "max_flow_mag", "wall_seconds", "neg_min_flow"]
_N_SUM = 4   # first 4 are SUM-reduced, the rest MAX-reduced


def reduce_stats(stats, group=None):
    """One all-reduce pair (SUM part, MAX part) on a 7-float vector: RCCL over xGMI when the process group's backend is
    "nccl", gloo on CPU.  No tensor data ever crosses GPUs."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return stats
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    v = torch.tensor([float(stats[k]) for k in STAT_NAMES], dtype=torch.float64, device=dev)
    s, m = v[:_N_SUM].clone(), v[_N_SUM:].clone()
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    out = torch.cat([s, m]).cpu().tolist()
    return dict(zip(STAT_NAMES, out))


def pair_stats(flow_mix, fill_mask):
    mag = torch.linalg.vector_norm(flow_mix.reshape(-1, 2), dim=1)
    return dict(pairs=1, sum_flow_mag=float(mag.sum()), hole_px=float(fill_mask.sum()), kernel_seconds=0.0,
                max_flow_mag=float(mag.max()), wall_seconds=0.0, neg_min_flow=float(-flow_mix.min()))


class DeviceStats:
    """The per-pair statistics stay ON the device (one mpf_pair_stats launch per pair into its own slot, no host
    synchronisation); result() reduces the slots and reads them back once."""

    CHUNK = 1024

    def __init__(self, device):
        self.device = torch.device(device)
        self.chunks, self.n = [], 0

    def add(self, flow_mix, fill_mask):
        from . import ops
        if self.n % self.CHUNK == 0:
            # empty, not zeros: rows are written whole by mpf_pair_stats, possibly from several streams, and only written rows are read
            self.chunks.append(torch.empty((self.CHUNK, ops.PAIR_STATS_SLICES, 4), dtype=torch.float64, device=self.device))
        ops.pair_stats(flow_mix, fill_mask, self.chunks[-1][self.n % self.CHUNK])
        self.n += 1

    def result(self, pairs):
        out = empty_stats()
        if self.n:
            rows = torch.cat(self.chunks)[:self.n].reshape(-1, 4)
            s, m = rows[:, :2].sum(0).cpu().tolist(), rows[:, 2:].max(0).values.cpu().tolist()
            out.update(pairs=pairs, sum_flow_mag=s[0], hole_px=s[1], max_flow_mag=m[0], neg_min_flow=m[1])
        return out


def merge_stats(a, b):
    out = {}
    for i, k in enumerate(STAT_NAMES):
        out[k] = (a[k] + b[k]) if i < _N_SUM else max(a[k], b[k])
    return out


def empty_stats():
    return dict(pairs=0, sum_flow_mag=0.0, hole_px=0.0, kernel_seconds=0.0, max_flow_mag=0.0, wall_seconds=0.0,
                neg_min_flow=-float("inf"))
