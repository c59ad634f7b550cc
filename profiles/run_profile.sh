#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/run_profile.sh <tag>
# Collects (1) rocprofv3 --kernel-trace --stats of the bench command, (2) separate PMC passes for FETCH_SIZE and
# WRITE_SIZE (never combined with tracing flags), all under gpurun_out/<tag>/ ; profiles/summarize.py turns them into
# the committed summaries.
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the driver's command with fewer pairs per step (the profiler serialises and records every launch); same kernels, same launch form
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --pairs-per-step 16 --no-cpu-baseline --no-sub"
cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
# PMC passes: the pipelined run (k_pair_overlap) followed by the serial one (k_src_blend_flow = the calibration kernel, k_warp_composite_views)
PMC="python $REPO/bench.py --steps 2 --warmup 1 --pairs-per-step 8 --no-cpu-baseline --no-sub"
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $PMC > $OUT/pmc_fetch.log 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $PMC > $OUT/pmc_write.log 2>&1
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch/serial -o bench -- $PMC --pipeline serial > $OUT/pmc_fetch_serial.log 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write/serial -o bench -- $PMC --pipeline serial > $OUT/pmc_write_serial.log 2>&1
cd $REPO
find $OUT -type f | head -50
python profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
python profiles/update_traffic.py $OUT > $OUT/traffic.log 2>&1
cat $OUT/traffic.log
