#!/bin/bash
# Run ON THE GPU BOX: PMC passes (no tracing flags) over tools/bench_overlap.py - the serial pair (k_src_blend_flow, k_warp_composite_views)
# next to the heterogeneous-grid launch (k_pair_overlap).  usage: bash profiles/pmc_pair_overlap.sh <tag>
TAG=${1:-pmc_ovl}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/bench_overlap.py --steps 2 --images 2 --pmc"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o b -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/ta -o b -- $CMD > $OUT/ta.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o b -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o b -- $CMD > $OUT/write.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if any(t in kn for t in ("warp_composite_views", "k_pair_overlap", "k_src_blend_flow")):
            vals[(kn.split("(")[0].split("<")[0][-28:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(vals.items()):
        print("%-30s %-34s n=%3d mean %.5g" % (k, c, len(v), sum(v) / len(v)))
PY
tail -3 $OUT/*.log | grep -i -E "error|fail" | head
