#!/bin/bash
# the generator as EIGHT ranks at the real shape (64 x 384 x 1280) on ONE GPU - gloo instead of RCCL, the device time-shared, so the
# pairs/s of this run mean nothing; what it shows is the HOST side of 8 ranks on one node: every rank's decode / submit / writer threads on the node's cores at
# once, `--writers` at its 8-rank default, the rendezvous, the mask-maximum broadcast, i % 8 sharding and the statistics all-reduce at full size.
# usage (on the GPU box, repo root): bash tools/host_share_8ranks.sh [n_images]
N=${1:-64}
TMP=$(mktemp -d)
python - "$TMP" "$N" <<'PY'
import os, sys
import numpy as np
from PIL import Image
tmp, n = sys.argv[1], int(sys.argv[2])
base = os.path.join(tmp, "data")
for d in ("images", "disps", "masks"):
    os.makedirs(os.path.join(base, d))
rs = np.random.RandomState(0)
yy, xx = np.mgrid[0:375, 0:1242]
for i in range(n):
    img = (np.clip(0.5 + 0.25 * np.sin(xx / (17.0 + i)) + 0.25 * np.cos(yy / 23.0) + 0.05 * rs.randn(375, 1242), 0, 1) * 255).astype(np.uint8)
    Image.fromarray(np.stack([img, np.roll(img, 7, 1), np.roll(img, 13, 0)], -1)).save(os.path.join(base, "images", "%04d.png" % i))
    Image.fromarray((255 * (0.1 + 0.8 * yy / 375)).astype(np.uint8)).save(os.path.join(base, "disps", "%04d.png" % i))
    m = np.zeros((375, 1242), np.uint8)
    m[150:300, 300:600] = 1
    m[200:330, 800:1000] = 2
    Image.fromarray(m).save(os.path.join(base, "masks", "%04d.png" % i))
PY
echo "== 8 ranks, gloo, one device (MPIFLOW_FORCE_DEVICE=0), $N images x 5 pairs at 64 x 384 x 1280"
T0=$(date +%s.%N)
env MPIFLOW_DIST_BACKEND=gloo MPIFLOW_FORCE_DEVICE=0 python gen_3dphoto_dynamic.py --gpus 8 --base $TMP/data --out $TMP/out8 --repeat 5 \
    --mpi-from model --ckpt_path random:0 --model-engine hip --inpaint builtin > $TMP/log8.txt 2>&1
echo "exit status $? after $(python -c "import time;print('%.1f' % (time.time() - $T0))") s"
grep -E "writers:|steady state|start-up|pairs |Error|error|Traceback" $TMP/log8.txt | head -20
tail -5 $TMP/log8.txt
echo "== 1 rank, same set (for the file comparison)"
python gen_3dphoto_dynamic.py --base $TMP/data --out $TMP/out1 --repeat 5 --mpi-from model --ckpt_path random:0 --model-engine hip --inpaint builtin 2>&1 | grep -E "writers:|steady state|pairs "
python - "$TMP" <<'PY'
import hashlib, os, sys
tmp = sys.argv[1]
bad = n = 0
for d in ("flows", "dst_images", "src_images"):
    for f in sorted(os.listdir(os.path.join(tmp, "out1", d))):
        a = hashlib.sha256(open(os.path.join(tmp, "out1", d, f), "rb").read()).hexdigest()
        p = os.path.join(tmp, "out8", d, f)
        b = hashlib.sha256(open(p, "rb").read()).hexdigest() if os.path.exists(p) else None
        n += 1
        bad += a != b
print("files of the 8-rank run vs the 1-rank run: %d compared, %d differ" % (n, bad))
PY
rm -rf $TMP
