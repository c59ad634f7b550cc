#!/bin/bash
# Run ON THE GPU BOX: PMC passes (no tracing flags) over the producer-engine forward.  usage: bash profiles/pmc_engine.sh <tag>
TAG=${1:-pmce}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/run_engine_once.py"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "Counter_Name:\s*[A-Z_a-z0-9]*MFMA[A-Za-z_0-9]*" | sort -u > $OUT/mfma_counters.txt
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/inst -o b -- $CMD > $OUT/inst.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $OUT/wait -o b -- $CMD > $OUT/wait.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/mfma -o b -- $CMD > $OUT/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/mem -o b -- $CMD > $OUT/mem.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections, re
rows = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_conv3x3" not in r["Kernel_Name"]:
            continue
        m = re.search(r"k_conv3x3<([^>]*)>", r["Kernel_Name"])
        key = (m.group(1).replace(" ", ""), r["Grid_Size"] if "Grid_Size" in r else "")
        rows.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in rows.items():
    print("k_conv3x3<%s> grid %s" % key)
    for c, v in sorted(d.items()):
        print("    %-34s n=%2d mean %.5g" % (c, len(v), sum(v) / len(v)))
PY
grep -i -E "error|fail|invalid" $OUT/*.log | head
