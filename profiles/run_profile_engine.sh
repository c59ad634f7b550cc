#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/run_profile_engine.sh <tag>
# rocprofv3 --kernel-trace --stats of (1) two forwards of the producer engine at 64x384x1280 and (2) the end-to-end generator
# on a 24-image synthetic KITTI-shaped dataset; per-kernel summaries are printed by profiles/summarize_kernels.py.
TAG=${1:-engine}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python $REPO/tools/run_engine_once.py > /dev/null 2>&1          # MIOpen picks its kernels outside the trace
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/engine -o e -- python $REPO/tools/run_engine_once.py > $OUT/engine.log 2>&1
cd $REPO
python tools/bench_engine.py > $OUT/bench_engine.txt 2>&1
python tools/bench_cli.py 160 > $OUT/bench_cli.txt 2>&1
python profiles/summarize_kernels.py $OUT/engine > $OUT/engine_kernels.txt 2>&1
tail -30 $OUT/engine_kernels.txt
