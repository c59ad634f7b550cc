#!/bin/bash
# kernel trace of the generator CLI (40 images x 5 pairs), summarised per kernel
REPO=$(pwd); OUT=$REPO/gpurun_out/cli_trace; mkdir -p $OUT; export TMPDIR=/tmp
python - <<PY
import os, numpy as np
from PIL import Image
base="/tmp/clidata"
for d in ("images","disps","masks"): os.makedirs(os.path.join(base,d), exist_ok=True)
rs=np.random.RandomState(0); yy,xx=np.mgrid[0:375,0:1242]
for i in range(40):
    img=(np.clip(0.5+0.25*np.sin(xx/(17.0+i))+0.25*np.cos(yy/23.0)+0.05*rs.randn(375,1242),0,1)*255).astype(np.uint8)
    Image.fromarray(np.stack([img,np.roll(img,7,1),np.roll(img,13,0)],-1)).save(os.path.join(base,"images","%04d.png"%i))
    Image.fromarray((255*(0.1+0.8*yy/375)).astype(np.uint8)).save(os.path.join(base,"disps","%04d.png"%i))
    m=np.zeros((375,1242),np.uint8); m[150:300,300:600]=1; m[200:330,800:1000]=2
    Image.fromarray(m).save(os.path.join(base,"masks","%04d.png"%i))
PY
python $REPO/gen_3dphoto_dynamic.py --base /tmp/clidata --out /tmp/cliout0 --repeat 5 --mpi-from model --ckpt_path random:0 --inpaint builtin > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o c -- python $REPO/gen_3dphoto_dynamic.py --base /tmp/clidata --out /tmp/cliout1 --repeat 5 --mpi-from model --ckpt_path random:0 --inpaint builtin > $OUT/run.log 2>&1
cd $REPO
python profiles/summarize_kernels.py $OUT/t | head -40
tail -2 $OUT/run.log
