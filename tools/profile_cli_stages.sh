set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, subprocess, sys, tempfile, numpy as np
from PIL import Image
tmp = tempfile.mkdtemp(prefix="mpf_cli_")
base = os.path.join(tmp, "data")
for d in ("images", "disps", "masks"):
    os.makedirs(os.path.join(base, d))
rs = np.random.RandomState(0)
yy, xx = np.mgrid[0:375, 0:1242]
for i in range(120):
    img = (np.clip(0.5 + 0.25 * np.sin(xx / (17.0 + i)) + 0.25 * np.cos(yy / 23.0) + 0.05 * rs.randn(375, 1242), 0, 1) * 255).astype(np.uint8)
    Image.fromarray(np.stack([img, np.roll(img, 7, 1), np.roll(img, 13, 0)], -1)).save(os.path.join(base, "images", "%04d.png" % i))
    Image.fromarray((255 * (0.1 + 0.8 * yy / 375)).astype(np.uint8)).save(os.path.join(base, "disps", "%04d.png" % i))
    m = np.zeros((375, 1242), np.uint8); m[150:300, 300:600] = 1; m[200:330, 800:1000] = 2
    Image.fromarray(m).save(os.path.join(base, "masks", "%04d.png" % i))
import json
variants = json.loads(os.environ.get("VARIANTS", "null")) or [[{"MPIFLOW_PROFILE": "host"}, "builtin"], [{"MPIFLOW_PROFILE": "host"}, "none"]]
for env, fill in variants:
    r = subprocess.run([sys.executable, "gen_3dphoto_dynamic.py", "--base", base, "--out", os.path.join(tmp, "o"), "--ckpt_path", "random:0",
                        "--inpaint", fill], capture_output=True, text=True, env=dict(os.environ, **env))
    print(env, fill, "\n", "\n".join(l for l in r.stdout.splitlines() if l.startswith("  ") or l.startswith("pairs") or l.startswith("steady")), r.stderr[-500:] if r.returncode else "")
PY
