"""Wire formats either side of the path: Middlebury .flo and BGR PNG frames.

.flo layout as the reference writes it (write_flow.py:74-103) and RAFT's loader reads it (write_flow.py:14-33):
float32 magic 202021.25, int32 width, int32 height, then height*width interleaved (u, v) float32, row-major,
little-endian."""
import numpy as np

FLO_MAGIC = np.float32(202021.25)


def write_flo(path, flow_HW2):
    flow = np.ascontiguousarray(flow_HW2, dtype=np.float32)
    assert flow.ndim == 3 and flow.shape[2] == 2
    h, w = flow.shape[:2]
    with open(path, "wb") as f:
        f.write(np.array([FLO_MAGIC], np.float32).tobytes())
        f.write(np.array([w, h], np.int32).tobytes())
        f.write(flow.tobytes())


def read_flo(path):
    with open(path, "rb") as f:
        magic = np.frombuffer(f.read(4), np.float32)[0]
        if magic != FLO_MAGIC:
            raise ValueError("Magic number incorrect. Invalid .flo file")
        w, h = np.frombuffer(f.read(8), np.int32)
        data = np.frombuffer(f.read(int(w) * int(h) * 8), np.float32)
    return data.reshape(int(h), int(w), 2).copy()


# ---- PNG: 8-bit RGB, one IDAT, filter "Up" on every row, deflate level 1 with the RLE strategy --------------------------------
# The reference writes its frames with cv2.imwrite at OpenCV's defaults (gen_3dphoto_dynamic_v2.py:121-122), which are
# tuned for speed (fast deflate, RLE strategy) rather than size; readers only see the pixels.  Pillow's encoder (level 6,
# adaptive filters, GIL held between rows) costs ~60-100 ms per 384x1280 frame and serialises the writer threads; this one
# is ~5x faster, spends its time in zlib (GIL released) and accepts scanlines that were filtered on the GPU
# (mpf_png_filter_up), so the writer thread only deflates and writes.
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _png_chunk(tag, data):
    import struct
    import zlib
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(data, zlib.crc32(tag)) & 0xFFFFFFFF)


def png_from_scanlines(scanlines_H_1p3W_u8, level=1):
    """Filtered scanlines u8 [H, 1+3W] (filter byte + RGB row, as mpf_png_filter_up / filter_up_rgb produce) -> PNG bytes"""
    import struct
    import zlib
    sc = np.ascontiguousarray(scanlines_H_1p3W_u8, dtype=np.uint8)
    h, w = sc.shape[0], (sc.shape[1] - 1) // 3
    assert sc.ndim == 2 and sc.shape[1] == 3 * w + 1
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, zlib.Z_RLE)
    idat = co.compress(sc.data) + co.flush()
    return _PNG_SIG + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + _png_chunk(b"IDAT", idat) + _png_chunk(b"IEND", b"")


def filter_up_rgb(img_HW3_rgb_u8):
    """Host version of mpf_png_filter_up for RGB input: [H,W,3] u8 -> scanlines [H, 1+3W]"""
    img = np.ascontiguousarray(img_HW3_rgb_u8, dtype=np.uint8)
    h, w, _ = img.shape
    flat = img.reshape(h, 3 * w)
    sc = np.empty((h, 3 * w + 1), np.uint8)
    sc[:, 0] = 2
    sc[0, 1:] = flat[0]
    np.subtract(flat[1:], flat[:-1], out=sc[1:, 1:])
    return sc


def write_png_bgr(path, img_HW3_bgr_u8, level=1):
    """cv2.imwrite semantics: the array is BGR, the file stores RGB."""
    data = png_from_scanlines(filter_up_rgb(np.asarray(img_HW3_bgr_u8)[:, :, ::-1]), level)
    with open(path, "wb") as f:
        f.write(data)


def write_bytes(path, data):
    with open(path, "wb") as f:
        f.write(data)


# ---- decoding a disparity file the way cv2.imread(path, 0) does (reference utils/utils.py:42-43) -----------------------------
# cv2.imread(path, cv2.IMREAD_GRAYSCALE) always returns 8-bit single-channel data:
#   8-bit grey           the stored bytes
#   16-bit grey          the HIGH byte (libpng png_set_strip_16: value >> 8) - MiDaS' default output, the tool the reference's
#                        README recommends.  (PIL's .convert("L") SATURATES 16-bit data instead: everything above 255 -> 255.)
#   colour PNG           libpng's png_set_rgb_to_gray fixed point: (R*9798 + G*19235 + B*3735 + 16384) >> 15   [coefficients
#                        0.299 / 0.587 scaled by 32768, blue = the remainder]
#   colour JPEG / other  OpenCV's BGR2GRAY fixed point: (R*4899 + G*9617 + B*1868 + 8192) >> 14
#   alpha is dropped, palettes are expanded first, 1-bit images become 0 / 255.
# OpenCV is third-party and absent here: this follows its documented behaviour and is compared with the real cv2.imread by
# tests/test_host_logic.py whenever OpenCV is importable (parity of this decoder is otherwise unpinned).
def grey_like_cv2_imread(pil_image, is_png=True):
    im = pil_image
    if im.mode in ("I;16", "I;16L", "I;16B", "I;16N"):
        return (np.asarray(im).astype(np.uint16) >> 8).astype(np.uint8)
    if im.mode == "I":                                   # 32-bit container PIL uses for 16-bit PNGs on some versions
        return ((np.asarray(im).astype(np.int64) & 0xFFFF) >> 8).astype(np.uint8)
    if im.mode == "F":
        raise ValueError("floating-point image files are not a cv2.imread(path, 0) input")
    if im.mode == "1":
        return np.where(np.asarray(im), 255, 0).astype(np.uint8)
    if im.mode in ("L",):
        return np.asarray(im).copy()
    if im.mode == "LA":
        return np.asarray(im)[..., 0].copy()
    rgb = np.asarray(im.convert("RGB")).astype(np.uint32)     # P, RGBA, CMYK, YCbCr ... -> RGB first
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    if is_png:
        return ((r * 9798 + g * 19235 + b * 3735 + 16384) >> 15).astype(np.uint8)
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def read_disparity_u8(path):
    """cv2.imread(path, 0) -> u8 [h,w]"""
    from PIL import Image
    im = Image.open(path)
    return grey_like_cv2_imread(im, is_png=(im.format or "").upper() == "PNG")


class AsyncWriter:
    """Bounded pool of writer threads so that PNG encoding (zlib, ~20 ms per 640x960 frame) and file I/O overlap the GPU
    instead of capping pairs/s.  Jobs are plain host arrays; submit() blocks when `max_pending` jobs
    are in flight (bounds host memory); close() waits for everything and re-raises the first writer exception."""

    def __init__(self, threads=8, max_pending=64):
        import concurrent.futures
        import threading
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=threads)
        self._slots = threading.Semaphore(max_pending)
        self._futures = []

    def _run(self, fn, args):
        try:
            fn(*args)
        finally:
            self._slots.release()

    def submit(self, fn, *args):
        self._slots.acquire()
        self._futures.append(self._pool.submit(self._run, fn, args))
        if len(self._futures) > 4096:
            self._futures = [f for f in self._futures if not f.done() or f.exception() is not None]

    def flo(self, path, flow_HW2):
        self.submit(write_flo, path, flow_HW2)

    def png_bgr(self, path, img):
        self.submit(write_png_bgr, path, img)

    def close(self):
        self._pool.shutdown(wait=True)
        for f in self._futures:
            if f.exception() is not None:
                raise f.exception()


class OutputRing:
    """Pairs leave the GPU without ever stalling the submitting thread.

    A ring of pinned host slots (flow [H,W,2] f32 + PNG scanlines of the rendered frame); submit() enqueues the
    device->host copies on the current stream, records an event and hands the slot to a writer thread, which waits for the
    event (GIL released), deflates, writes the files and returns the slot.  submit() only blocks when every slot is in
    flight (back-pressure).  The source frame of an image is the same for all its pairs: submit_source() encodes it once
    and writes it under every pair's name."""

    def __init__(self, H, W, device, slots=16, threads=8, png_level=1, host_fill=None):
        """host_fill: None, or a function (frame_bgr u8 [H,W,3], hole u8 [H,W]) -> filled frame, run on the writer thread before
        the PNG is encoded - the reference's cv2.inpaint step (utils/utils.py:284-286), which is sequential host work by nature
        and is overlapped here with the GPU render of the following pairs (submit_pair_fill)."""
        import concurrent.futures
        import queue
        import torch
        self.H, self.W, self.level, self._host_fill = H, W, png_level, host_fill
        # slots are page-locked on demand, up to `slots` of them: pinning all 64 up front cost 0.4-0.7 s of start-up (7 MB each), and a run
        # whose writers keep up never needs more than a handful
        self._free = queue.Queue()
        self._max_slots, self._n_slots, self._slot_lock = slots, 0, __import__("threading").Lock()
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=threads)
        self._futures = []
        self._torch = torch
        # cumulative seconds the writer threads spent per stage (stage_report()): which host stage would fall behind when the ranks of a node share its cores
        self.threads, self._stage, self._stage_lock, self._t_open = threads, {}, __import__("threading").Lock(), __import__("time").perf_counter()
        self.backpressure_seconds = 0.0

    def _lap(self, name, t0):
        import time
        t1 = time.perf_counter()
        with self._stage_lock:
            c = self._stage.setdefault(name, [0.0, 0])
            c[0] += t1 - t0
            c[1] += 1
        return t1

    def stage_report(self):
        """-> dict(stage -> (seconds, calls)), threads, busy share of the writer pool since it was opened, seconds submit() spent waiting for a free slot"""
        import time
        with self._stage_lock:
            st = {k: tuple(v) for k, v in self._stage.items()}
        wall = time.perf_counter() - self._t_open
        busy = sum(v[0] for k, v in st.items() if k != "wait for the GPU (event)")
        return dict(stages=st, threads=self.threads, wall_seconds=wall, busy_share=busy / max(1e-9, wall * self.threads), backpressure_seconds=self.backpressure_seconds)

    def _new_slot(self):
        torch, H, W = self._torch, self.H, self.W
        slot = dict(scan=torch.empty((H, 3 * W + 1), dtype=torch.uint8).pin_memory())
        if self._host_fill is None:
            slot.update(flow=torch.empty((H, W, 2), dtype=torch.float32).pin_memory())
        else:
            # flow | frame | hole in ONE page-locked buffer with the layout of ops.pair_slab: a pair rendered into such a slab leaves the GPU in one
            # device-to-host copy (3 per pair before: on this ROCm build every copy to pinned memory is a blit kernel that parks a workgroup per CU)
            n = H * W
            slab = torch.empty(12 * n, dtype=torch.uint8).pin_memory()
            slot.update(slab=slab, flow=slab[:8 * n].view(torch.float32).view(H, W, 2), frame=slab[8 * n:11 * n].view(H, W, 3), hole=slab[11 * n:].view(H, W))
        return slot

    def _get_slot(self):
        """A free slot; a new one while fewer than the maximum exist; else wait for a writer to hand one back (back-pressure)."""
        import queue
        try:
            return self._free.get_nowait()
        except queue.Empty:
            pass
        with self._slot_lock:
            grow = self._n_slots < self._max_slots
            if grow:
                self._n_slots += 1
        if grow:
            return self._new_slot()
        import time
        t0 = time.perf_counter()
        slot = self._free.get()                                      # every slot is in flight: the writers are behind (back-pressure on the submitting thread)
        self.backpressure_seconds += time.perf_counter() - t0
        return slot

    def _finish(self, slot, event, flo_path, png_paths, fill=False):
        import time
        try:
            t = time.perf_counter()
            event.synchronize()
            t = self._lap("wait for the GPU (event)", t)
            if flo_path is not None:
                write_flo(flo_path, slot["flow"].numpy())
                t = self._lap(".flo file", t)
            if png_paths:
                if fill:
                    frame = self._host_fill(slot["frame"].numpy(), slot["hole"].numpy())
                    t = self._lap("hole fill (NS)", t)
                    scan = filter_up_rgb(np.asarray(frame)[:, :, ::-1])               # cv2.imwrite: BGR array -> RGB file
                    t = self._lap("PNG Up filter (host)", t)
                else:
                    scan = slot["scan"].numpy()
                data = png_from_scanlines(scan, self.level)
                t = self._lap("PNG deflate", t)
                for p in png_paths:
                    write_bytes(p, data)
                t = self._lap("PNG file", t)
        finally:
            self._free.put(slot)

    def submit_pair_fill(self, flow_HW2_dev, frame_bgr_dev, hole_dev, flo_path, png_path, slab=None):
        """Like submit_pair, but the frame leaves the GPU unfilled together with its hole mask and the writer thread runs
        `host_fill` on it before encoding.  slab: the device buffer the three tensors are views of (ops.pair_slab) - then ONE copy."""
        assert self._host_fill is not None
        slot = self._get_slot()
        if slab is not None and slab.numel() == slot["slab"].numel():
            slot["slab"].copy_(slab, non_blocking=True)
        else:
            slot["flow"].copy_(flow_HW2_dev, non_blocking=True)
            slot["frame"].copy_(frame_bgr_dev, non_blocking=True)
            slot["hole"].copy_(hole_dev, non_blocking=True)
        ev = self._torch.cuda.Event()
        ev.record()
        self._track(self._pool.submit(self._finish, slot, ev, flo_path, [png_path], True))

    def _track(self, fut):
        self._futures.append(fut)
        if len(self._futures) > 1024:
            done = [f for f in self._futures if f.done()]
            for f in done:
                f.result()
            self._futures = [f for f in self._futures if not f.done()]

    def _submit(self, flow_dev, scan_dev, flo_path, png_paths):
        slot = self._get_slot()
        if flow_dev is not None:
            slot["flow"].copy_(flow_dev, non_blocking=True)
        slot["scan"].copy_(scan_dev, non_blocking=True)
        ev = self._torch.cuda.Event()
        ev.record()
        self._track(self._pool.submit(self._finish, slot, ev, flo_path, png_paths))

    def submit_pair(self, flow_HW2_dev, frame_scanlines_dev, flo_path, png_path):
        self._submit(flow_HW2_dev, frame_scanlines_dev, flo_path, [png_path])

    def submit_source(self, src_scanlines_dev, png_paths):
        self._submit(None, src_scanlines_dev, None, list(png_paths))

    def close(self):
        self._pool.shutdown(wait=True)
        for f in self._futures:
            f.result()
        self._futures = []


class InputPrefetcher:
    """Decodes the three PNGs of the images THIS RANK OWNS on background threads, so the GPU never waits
    for a decoder.  Iterating yields, in listing order of `indices`, dicts
        {i, name, error: None | Exception, rgb_u8 [h,w,3], disp_u8 [h,w], ids_u8 [h,w]}
    - the uint8 arrays exactly as the reference's loaders see them before their `/255` (image: PIL RGB, utils/utils.py:35-39;
    disparity: cv2.imread(path, 0), :42-43; mask: PIL "L", gen_3dphoto_dynamic_v2.py:83) - as torch tensors in page-locked
    memory (pin=True) ready for an asynchronous upload; the float conversion and the resize happen on the GPU
    (mpf_prepare_inputs).  The three arrays of an image need NOT share one (h, w): each is resized on its own.  A file that cannot be decoded does not raise here: its item carries `error`, and the driver skips
    that image and carries on."""

    def __init__(self, names, img_dir, disp_dir, mask_dir, indices, depth=8, threads=4, pin=True):
        import collections
        import concurrent.futures
        self._names, self._dirs, self._todo, self._pin = list(names), (img_dir, disp_dir, mask_dir), list(indices), pin
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=threads)
        self._pending = collections.deque()
        self._next, self._depth = 0, depth
        self._fill()

    def _fill(self):
        while self._next < len(self._todo) and len(self._pending) < self._depth:
            self._pending.append(self._pool.submit(self._load, self._todo[self._next]))
            self._next += 1

    def _load(self, i):
        import os
        from PIL import Image
        img_dir, disp_dir, mask_dir = self._dirs
        n = self._names[i]
        try:
            ids = np.array(Image.open(os.path.join(mask_dir, n)).convert("L"))
            rgb = np.array(Image.open(os.path.join(img_dir, n)).convert("RGB"))
            disp = read_disparity_u8(os.path.join(disp_dir, n))
            # the three files may differ in size (e.g. a MiDaS/DPT disparity saved at network resolution): the reference resizes each
            # of them on its own to (height, width) (gen_3dphoto_dynamic_v2.py:82-89, :104-105), and so does mpf_prepare_inputs
            out = dict(i=i, name=n, error=None, rgb_u8=rgb, disp_u8=disp, ids_u8=ids)
            if self._pin:
                import torch
                for k in ("rgb_u8", "disp_u8", "ids_u8"):
                    out[k] = torch.from_numpy(np.ascontiguousarray(out[k])).pin_memory()
            return out
        except Exception as e:                      # noqa: BLE001 - reported per image by the driver
            return dict(i=i, name=n, error=e)

    def __iter__(self):
        try:
            while self._pending:
                item = self._pending.popleft().result()
                self._fill()
                yield item
        finally:
            self._pool.shutdown(wait=False, cancel_futures=True)


def mask_max_of_file(path):
    """np.array(Image.open(path).convert("L")).max() (gen_3dphoto_dynamic_v2.py:83, :101), or -1 when the file cannot be decoded"""
    from PIL import Image
    try:
        return int(np.array(Image.open(path).convert("L")).max())
    except Exception:                               # noqa: BLE001
        return -1
