#!/bin/bash
# Run ON THE GPU BOX: instruction / busy counters of the few-block conv kernels with one plane per workgroup (MPIFLOW_PW=...=1) and with the walking form.
# usage: bash profiles/pmc_engine_pw.sh <tag>
TAG=${1:-pmcpw}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/run_engine_once.py"
$CMD > /dev/null 2>&1
cd /tmp
F='--kernel-include-regex k_conv3x3'
for mode in one walk; do
  if [ $mode = one ]; then export MPIFLOW_PW="l8s=1,l9=1,up1_0=1,disp0=1,l7=1,up0_0=1"; else export MPIFLOW_PW="l7=4,up0_0=4"; fi
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES $F --output-format csv -d $OUT/$mode/inst -o b -- $CMD > $OUT/$mode.inst.log 2>&1
  timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE $F --output-format csv -d $OUT/$mode/wait -o b -- $CMD > $OUT/$mode.wait.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections, re
for mode in ("one", "walk"):
    rows = collections.OrderedDict()
    for f in sorted(glob.glob("$OUT/%s/**/*counter_collection.csv" % mode, recursive=True)):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_conv3x3<[^>]*>)", r["Kernel_Name"])
            if m:
                rows.setdefault((m.group(1).replace(" ", ""), r.get("Grid_Size", "")), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# MPIFLOW_PW mode: %s   (k_conv3x3<stride, ct, loader, epilogue, blocks, th, tw, wlds[, walk]>)" % mode)
    for key, d in rows.items():
        mean = lambda n: sum(d[n]) / len(d[n]) if n in d else float("nan")
        if key[0].split(",")[4] not in ("1", "2"):
            continue
        cyc = mean("GRBM_GUI_ACTIVE") / 8.0
        print("%-46s grid %-9s %7.1f us  VALU %.4g  MFMA %.4g  SALU %.4g  LDS %.4g  VMEM_RD %.4g  waves %.4g | VALU busy %4.1f %%  MFMA busy %4.1f %%  wait_any %.3g  wait_inst_any %.3g" % (
            key[0], key[1], cyc / 2400.0, mean("SQ_INSTS_VALU"), mean("SQ_INSTS_MFMA"), mean("SQ_INSTS_SALU"), mean("SQ_INSTS_LDS"), mean("SQ_INSTS_VMEM_RD"), mean("SQ_WAVES"),
            100.0 * mean("SQ_ACTIVE_INST_VALU") * 4.0 / (cyc * 1024.0), 100.0 * mean("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0), mean("SQ_WAIT_ANY"), mean("SQ_WAIT_INST_ANY")))
PY
