#!/bin/bash
# blocks per workgroup (MPIFLOW_NF: feature blocks, NB = 2 nf) and channels per tap (MPIFLOW_CT) of the phase-decomposed layers, one process per setting
run() { MPIFLOW_NF="$1" MPIFLOW_CT="$2" python tools/bench_engine.py 2>/dev/null | awk -v t="NF=$1 CT=$2" '/up1_[1-4]/{printf "%s=%s ", $(NF-2), $6} /hipGraph/{print " | graph", $(NF-1), "ms   [" t "]"}'; }
run "" ""
run "up1_3=2" ""
run "up1_3=1" ""
run "up1_2=1" ""
run "up1_3=2,up1_2=1" ""
run "" "up1_4=16"
run "up1_4=3" "up1_4=16"
run "" ""
