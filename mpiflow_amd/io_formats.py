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


def write_png_bgr(path, img_HW3_bgr_u8):
    """cv2.imwrite semantics: the array is BGR, the file stores RGB."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(np.asarray(img_HW3_bgr_u8)[:, :, ::-1])).save(path)


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
