#!/bin/bash
# Run ON THE GPU BOX (every pass under its own `timeout`): planar Stage B, workgroup-wide LDS footprints (planar_lds 1) against wave-private ones (planar_lds 2, no barrier in the plane loop):
# instruction and wait counters per dispatch (PMC passes only, no tracing flags).  usage: bash profiles/pmc_planar_wave.sh <tag>
TAG=${1:-pmcpw}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for lds in 1 2; do
  CMD="python $REPO/tools/bench_stage_b.py --layout 0 --mask 1 --aux 0 --planar-lds $lds --variants 1 --launches 3 --rounds 1"
  F='--kernel-include-regex k_warp_composite_planar'
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS $F --output-format csv -d $OUT/a$lds -o b -- $CMD > $OUT/a$lds.log 2>&1
  timeout 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY GRBM_GUI_ACTIVE TA_BUSY_avr $F --output-format csv -d $OUT/b$lds -o b -- $CMD > $OUT/b$lds.log 2>&1
  timeout 150 rocprofv3 --pmc FETCH_SIZE $F --output-format csv -d $OUT/c$lds -o b -- $CMD > $OUT/c$lds.log 2>&1   # alone: with WRITE_SIZE + TCC_* in the same pass this command did not finish in 25 min
done
# the sharing experiment of tools/bench_shared_views.py: FETCH_SIZE per launch form
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_warp_composite_views --output-format csv -d $OUT/shared -o b -- python $REPO/tools/bench_shared_views.py --launches 2 --rounds 1 > $OUT/shared.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections, re
for tag in ("a1", "b1", "c1", "a2", "b2", "c2"):
    rows = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            rows[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(rows.items()):
        print("planar_lds=%s  %-50s %-24s %.4g" % (tag[1], k, c, sum(v) / len(v)))
print("# sharing experiment: dispatches in launch order (per view_shift: pair, cam twice, dyn twice, one view cam, one view dyn; warm-up + 2 timed each)")
rows = []
for f in glob.glob("$OUT/shared/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
by = collections.defaultdict(dict)
for d, c, v in rows:
    by[d][c] = v
for d in sorted(by):
    print("dispatch %4d  " % d + "  ".join("%s=%.4g" % kv for kv in sorted(by[d].items())))
PY
