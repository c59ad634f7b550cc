#!/usr/bin/env python3
"""Record what the REAL OpenCV returns for the reference's three cv2 calls on committed inputs -> tests/golden/cv2_golden.npz.

    python tests/golden/make_cv2_golden.py [--out PATH]          (wherever `import cv2` works; needs nothing else but numpy)

The reference's calls (rows A13 / N2 of SURVEY.md §8):
    cv2.inpaint(frame_mix, fill_mask, 3, cv2.INPAINT_NS)         utils/utils.py:284-286
    cv2.inpaint(im1_raw, 1 - H, 3, cv2.INPAINT_TELEA)            moving_obj.py:162
    cv2.dilate(M, np.ones((3, 3)))                               moving_obj.py:144-145
OpenCV is third-party, not vendored in the reference and absent from the build image, so those rows are "parity unpinned" until
this script has run somewhere.  Inputs are committed fixtures only (this file reads nothing outside tests/golden/):
  * frame_mix / fill_mask exactly as the reference handed them to cv2.inpaint (recorded from the reference by make_golden.py):
    tiny_white, tiny_smooth, odd_s20, odd_s5, s1;
  * the forward-warped frame and the hole mask 1 - H the reference handed to the Telea call: fwarp_small (warped[..., :3], inpaint_mask),
    and its collision mask M = warped[..., 4] for cv2.dilate;
  * inpaint_reading_exhibit.npz: two inputs on which the two possible readings of OpenCV's unqualified sqrt() / fabs() calls
    (float overloads vs double functions, oracle/oracle_inpaint.c) give DIFFERENT bytes - the outputs recorded here decide it.
Without cv2 the script can still (re)generate the exhibit inputs from the oracle: --exhibit-only.
tests/test_inpaint.py uses cv2_golden.npz when it exists; until then the cv2 comparisons skip and the rows stay unpinned."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PAIR_FIXTURES = ["tiny_white", "tiny_smooth", "odd_s20", "odd_s5", "s1"]


def exhibit_inputs():
    """The deterministic inputs of the reading exhibit (numpy RandomState draws, no oracle needed): (name, img, mask, method)"""
    out = []
    for name, seed, method in (("ns", 1085, 0), ("telea", 1094, 1)):
        rs = np.random.RandomState(seed)
        H, W = 48, 64
        img = (rs.rand(H, W, 3) * 256).astype(np.uint8)
        mask = (rs.rand(H, W) < 0.35).astype(np.uint8)
        mask[10:30, 20:40] = 1
        out.append((name, img, mask, method))
    return out


def make_exhibit(path):
    """Both readings' outputs by the oracle (test infrastructure) next to the inputs; asserts that they differ."""
    sys.path.insert(0, ROOT)
    from oracle import mpi_oracle as orc
    rec = {}
    for name, img, mask, method in exhibit_inputs():
        a, b = orc.inpaint(img, mask, 3, method, reading=0), orc.inpaint(img, mask, 3, method, reading=1)
        assert (a != b).any(), "exhibit %s does not discriminate the readings any more" % name
        rec.update({name + "_img": img, name + "_mask": mask, name + "_float_reading": a, name + "_double_reading": b,
                    name + "_bytes_that_differ": np.int64((a != b).sum())})
    np.savez_compressed(path, **rec)
    print("wrote", path, {k: int(v) for k, v in rec.items() if k.endswith("differ")})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "cv2_golden.npz"))
    ap.add_argument("--exhibit-only", action="store_true", help="only (re)generate inpaint_reading_exhibit.npz from the oracle")
    a = ap.parse_args()
    if a.exhibit_only:
        make_exhibit(os.path.join(HERE, "inpaint_reading_exhibit.npz"))
        return 0
    try:
        import cv2
    except Exception as e:                                  # noqa: BLE001
        print("make_cv2_golden: OpenCV is not importable here (%r): nothing recorded, rows A13 / N2 stay parity-unpinned" % (e,))
        return 3
    rec = {"cv2_version": np.array(cv2.__version__), "numpy_version": np.array(np.__version__)}
    for name in PAIR_FIXTURES:
        g = np.load(os.path.join(HERE, name + ".npz"))
        rec[name + "_ns"] = cv2.inpaint(g["frame_mix"], g["fill_mask"], 3, cv2.INPAINT_NS)                   # utils/utils.py:284-286
        rec[name + "_telea"] = cv2.inpaint(g["frame_mix"], g["fill_mask"], 3, cv2.INPAINT_TELEA)
    g = np.load(os.path.join(HERE, "fwarp_small.npz"))
    frame = np.ascontiguousarray(g["warped"][..., :3])
    rec["fwarp_small_telea"] = cv2.inpaint(frame, g["inpaint_mask"], 3, cv2.INPAINT_TELEA)                   # moving_obj.py:162
    rec["fwarp_small_ns"] = cv2.inpaint(frame, g["inpaint_mask"], 3, cv2.INPAINT_NS)
    M = np.ascontiguousarray(g["warped"][..., 4])
    rec["fwarp_small_dilate"] = cv2.dilate(M, np.ones((3, 3)))                                               # moving_obj.py:144-145
    for name, img, mask, method in exhibit_inputs():
        rec["exhibit_" + name] = cv2.inpaint(img, mask, 3, cv2.INPAINT_TELEA if method == 1 else cv2.INPAINT_NS)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    np.savez_compressed(a.out, **rec)
    print("wrote %s with OpenCV %s: %d arrays" % (a.out, cv2.__version__, len(rec) - 2))
    return 0


if __name__ == "__main__":
    sys.exit(main())
