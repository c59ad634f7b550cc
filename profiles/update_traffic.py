#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE PMC passes of profiles/run_profile.sh into profiles/roofline_traffic.json.

usage (on the GPU box, after run_profile.sh <tag>):  python profiles/update_traffic.py gpurun_out/<tag>
HBM bytes per Stage B launch = FETCH_SIZE[KB] * 1024 * 2 + WRITE_SIZE[KB] * 1024: on gfx950 rocprofv3's FETCH_SIZE reports half the
bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); the factor is calibrated in the same run on k_src_blend_flow,
whose read and write volumes are known exactly.  The record carries the SHA-256 of the kernel source it was measured on;
bench.py reports `roofline.traffic` only while mpf_render.hip still has that digest."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def means(pattern, counter):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(root, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch, write = means("pmc_fetch", "FETCH_SIZE"), means("pmc_write", "WRITE_SIZE")
kb = [k for k in fetch if "k_warp_composite" in k]
kac = [k for k in fetch if "k_src_blend_flow" in k]
kp = [k for k in fetch if "k_pair_overlap" in k]
assert kb and kac, (list(fetch), list(write))
kb, kac = max(kb, key=lambda k: fetch[k]), max(kac, key=lambda k: fetch[k])
kp = max(kp, key=lambda k: fetch[k]) if kp else None
S, H, W = 64, 640, 960
N = H * W
views = 2 if "views" in kb else 1
alg_b = 16.0 * S * N * views
src = b"".join(open(os.path.join(repo, "mpiflow_amd", "csrc", f), "rb").read() for f in ("mpf_render.hip", "mpf_math.h"))
rec = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --steps 2` (profiles/run_profile.sh), this round",
    "kernel": kb, "views_per_launch": views,
    "FETCH_SIZE_KB_mean": fetch[kb], "WRITE_SIZE_KB_mean": write.get(kb),
    "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md); calibration on %s in the same run: FETCH_SIZE*2*1024 = %.1f MB vs %.1f MB algorithmic read, WRITE_SIZE*1024 = %.1f MB vs %.1f MB algorithmic write"
                  % (kac, fetch[kac] * 2 * 1024 / 1e6, (16.0 * S * N + 12.0 * N + 4.0 * N) / 1e6, write.get(kac, 0) * 1024 / 1e6, (16.0 * S * N + 16.0 * N + 2 * 32.0 * N + 3.0 * N) / 1e6),
    "stage_b_hbm_bytes_per_launch": fetch[kb] * 2 * 1024 + write.get(kb, 0) * 1024,
    "algorithmic_bytes_per_launch": alg_b,
    "kernel_source_sha256": hashlib.sha256(src).hexdigest(),
}
if kp:      # the heterogeneous-grid launch: Stage B of pair i (2 views) + Stage A+C of pair i + 1
    rec.update({"pair_kernel": kp, "pair_FETCH_SIZE_KB_mean": fetch[kp], "pair_WRITE_SIZE_KB_mean": write.get(kp),
                "pair_hbm_bytes_per_launch": fetch[kp] * 2 * 1024 + write.get(kp, 0) * 1024,
                "pair_algorithmic_bytes_per_launch": 60.0 * S * N + 28.0 * N,
                "pair_separate_kernels_hbm_bytes": fetch[kb] * 2 * 1024 + write.get(kb, 0) * 1024 + fetch[kac] * 2 * 1024 + write.get(kac, 0) * 1024})
json.dump(rec, open(os.path.join(root, "roofline_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
