#!/bin/bash
# Run ON THE GPU BOX: wave-lifetime / occupancy counters of the producer engine's kernels (no tracing flags).
# usage: bash profiles/pmc_engine_occupancy.sh <tag>
TAG=${1:-pmco}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/run_engine_once.py"
$CMD > /dev/null 2>&1
cd /tmp
F='--kernel-include-regex k_conv3x3'
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE $F --output-format csv -d $OUT/occ -o b -- $CMD > $OUT/occ.log 2>&1
timeout 500 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES GRBM_GUI_ACTIVE $F --output-format csv -d $OUT/occ2 -o b -- $CMD > $OUT/occ2.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections, re
rows = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_conv3x3<[^>]*>)", r["Kernel_Name"])
        if not m:
            continue
        key = (m.group(1).replace(" ", ""), r.get("Grid_Size", ""))
        rows.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in rows.items():
    print("%s  grid %s" % key)
    print("    " + "  ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in sorted(d.items())))
    if "SQ_WAVE_CYCLES" in d and "SQ_WAVES" in d and "GRBM_GUI_ACTIVE" in d:
        wc, w, g = (sum(d[n]) / len(d[n]) for n in ("SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"))
        print("    wave lifetime %.0f cycles (x4 if the counter ticks every 4) ; kernel %.0f cycles per XCD ; mean waves in flight per CU %.2f (x4?)" % (wc / w, g / 8, wc / (g / 8) / 256))
PY
tail -5 $OUT/occ.log $OUT/occ2.log
