#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE / TCC hit-miss of k_pair_overlap with the roles interleaved on every XCD (ovl_xcd_a=0) and
# partitioned by XCD (ovl_xcd_a=3): VERDICT r3 item 4.  usage: bash profiles/pmc_pair_xcd.sh <tag>
TAG=${1:-pmc_xcd}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for XA in 0 3; do
  CMD="python $REPO/bench.py --steps 2 --warmup 1 --pairs-per-step 8 --no-cpu-baseline --no-sub --no-moving-object --witness --tune ovl_xcd_a=$XA"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_xa$XA -o b -- $CMD > $OUT/fetch_xa$XA.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_xa$XA -o b -- $CMD > $OUT/write_xa$XA.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc_xa$XA -o b -- $CMD > $OUT/tcc_xa$XA.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_pair_overlap" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(vals.items()):
        print("%-40s k_pair_overlap %-14s n=%3d mean %.6g" % (f.split("/")[-3], c, len(v), sum(v) / len(v)))
PY
