#!/bin/bash
# planes-per-workgroup sweep of the few-block full-resolution layers (MPIFLOW_PW), same box, one process per setting: layer times + graph forward
for PW in "" "up1_0=8" "up1_0=16" "up1_0=2" "disp0=8" "l9=8" "l8s=8" "up0_0=8" "l7=8" "up1_0=8,disp0=8,l9=8"; do
  MPIFLOW_PW="$PW" python tools/bench_engine.py 2>/dev/null | awk -v pw="$PW" '/up1_0|disp0|l9 |l8s|up0_0|l7 /{printf "%s=%s ", $NF=="" ? $0 : $(NF-2), $6} /hipGraph/{print " | graph", $(NF-1), "ms   [MPIFLOW_PW=" pw "]"}'
done
