#!/bin/bash
# Run ON THE GPU BOX: L2 request counters of the conv kernels of one forward of the fast engine (how much the weight-fragment / map copies ask of L2).
# usage: bash profiles/pmc_engine_l2.sh <tag>
TAG=${1:-pmcl2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/run_engine_once.py"
$CMD > /dev/null 2>&1
cd /tmp
F='--kernel-include-regex k_conv3x3'
timeout 400 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE $F --output-format csv -d $OUT/l2 -o b -- $CMD > $OUT/l2.log 2>&1
timeout 400 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum $F --output-format csv -d $OUT/tcp -o b -- $CMD > $OUT/tcp.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections, re
rows = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_conv3x3<[^>]*>)", r["Kernel_Name"])
        if m:
            rows.setdefault((m.group(1).replace(" ", ""), r.get("Grid_Size", "")), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per dispatch (mean): L2 requests (TCC_REQ, 128-byte lines on gfx950), hits / misses, L1 -> L2 read / write requests; kernel time from GRBM_GUI_ACTIVE / 8 at 2.4 GHz")
for key, d in rows.items():
    mean = lambda n: sum(d[n]) / len(d[n]) if n in d else float("nan")
    us = mean("GRBM_GUI_ACTIVE") / 8.0 / 2400.0
    print("%-44s grid %-9s %7.1f us  TCC_REQ %.4g (%.1f GB at 128 B, %.1f TB/s)  hit %.4g  miss %.4g  TCP->TCC rd %.4g wr %.4g" % (
        key[0], key[1], us, mean("TCC_REQ_sum"), mean("TCC_REQ_sum") * 128 / 1e9, mean("TCC_REQ_sum") * 128 / 1e6 / us if us == us and us > 0 else float("nan"),
        mean("TCC_HIT_sum"), mean("TCC_MISS_sum"), mean("TCP_TCC_READ_REQ_sum"), mean("TCP_TCC_WRITE_REQ_sum")))
PY
