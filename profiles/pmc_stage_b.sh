#!/bin/bash
# Run ON THE GPU BOX: PMC passes (no tracing flags) over the Stage B micro-benchmark.  usage: bash profiles/pmc_stage_b.sh <tag> [variants]
TAG=${1:-pmcb}
VAR=${2:-1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/bench_stage_b.py --variants $VAR --rounds 1 --launches 4 --images 2"
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o b -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_BUSY_avr TA_BUSY_max TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/ta -o b -- $CMD > $OUT/ta.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/inst -o b -- $CMD > $OUT/inst.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/mem -o b -- $CMD > $OUT/mem.log 2>&1
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
