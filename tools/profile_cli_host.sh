# cProfile of the generator's submitting thread (host side): where do the ~11 ms per image go?  usage (GPU box): bash tools/profile_cli_host.sh [n_images]
set -e
cd $GRAFT_REPO_ROOT
N=${1:-60}
python - <<PY
import os, numpy as np
from PIL import Image
base = "/tmp/clihost"
for d in ("images", "disps", "masks"):
    os.makedirs(os.path.join(base, d), exist_ok=True)
rs = np.random.RandomState(0)
yy, xx = np.mgrid[0:375, 0:1242]
for i in range($N):
    img = (np.clip(0.5 + 0.25 * np.sin(xx / (17.0 + i)) + 0.25 * np.cos(yy / 23.0) + 0.05 * rs.randn(375, 1242), 0, 1) * 255).astype(np.uint8)
    Image.fromarray(np.stack([img, np.roll(img, 7, 1), np.roll(img, 13, 0)], -1)).save(os.path.join(base, "images", "%04d.png" % i))
    Image.fromarray((255 * (0.1 + 0.8 * yy / 375)).astype(np.uint8)).save(os.path.join(base, "disps", "%04d.png" % i))
    m = np.zeros((375, 1242), np.uint8); m[150:300, 300:600] = 1; m[200:330, 800:1000] = 2
    Image.fromarray(m).save(os.path.join(base, "masks", "%04d.png" % i))
PY
python -m cProfile -o /tmp/cli.prof gen_3dphoto_dynamic.py --base /tmp/clihost --out /tmp/clihost_out --ckpt_path random:0 --inpaint ${FILL:-none} | tail -3
python - <<PY
import pstats
p = pstats.Stats("/tmp/cli.prof")
p.sort_stats("tottime").print_stats(25)
p.print_callers("method 'to' of")
p.print_callers("pin_memory")
p.sort_stats("cumtime").print_stats("mpiflow_amd|gen_3dphoto", 45)
PY
