#!/bin/bash
# Run ON THE GPU BOX: PMC passes (no tracing flags) over tools/bench_stage_b_views.py.  usage: bash profiles/pmc_stage_b_views.sh <tag> <variants> <views>
TAG=${1:-pmcv}
VAR=${2:-1,20}
VIEWS=${3:-1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/bench_stage_b_views.py --variants $VAR --views $VIEWS --rounds 1 --launches 3 --images 2"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o b -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/lds -o b -- $CMD > $OUT/lds.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/ta -o b -- $CMD > $OUT/ta.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o b -- $CMD > $OUT/fetch.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "warp_composite" in r["Kernel_Name"]:
            vals[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(vals.items()):
        print("%-46s %-36s n=%3d mean %.4g" % (k, c, len(v), sum(v) / len(v)))
PY
tail -3 $OUT/*.log | grep -i -E "error|fail" | head
