#!/bin/bash
# Run ON THE GPU BOX: PMC passes (no tracing flags) over the producer-engine forward, restricted to the engine's own kernels
# (without the filter every MIOpen / torch kernel of the start-up is serialised under the counters and a pass takes minutes).
# usage: bash profiles/pmc_engine.sh <tag>
TAG=${1:-pmce}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/tools/run_engine_once.py"
$CMD > /dev/null 2>&1
cd /tmp
F='--kernel-include-regex k_conv3x3|k_plane_masks'
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES $F --output-format csv -d $OUT/inst -o b -- $CMD > $OUT/inst.log 2>&1
timeout 500 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE TA_BUSY_avr $F --output-format csv -d $OUT/wait -o b -- $CMD > $OUT/wait.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum $F --output-format csv -d $OUT/mem -o b -- $CMD > $OUT/mem.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections, re
rows = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_conv3x3(?:_up1?)?<[^>]*>|k_plane_masks)", r["Kernel_Name"])
        if not m:
            continue
        key = (m.group(1).replace(" ", ""), r.get("Grid_Size", ""))
        rows.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# k_conv3x3<stride, channels per tap, loader, epilogue, blocks, tile h, tile w, weights through LDS>; counters are per dispatch (mean)")
for key, d in rows.items():
    w = (sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])) if "SQ_WAVES" in d else 1
    print("%s  grid %s" % key)
    if "SQ_INSTS_VALU" in d:
        print("    per wave: " + "  ".join("%s %.0f" % (n[9:], sum(d[n]) / len(d[n]) / w) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_MFMA") if n in d))
    print("    " + "  ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in sorted(d.items())))
PY
# MFMA-busy share per layer: SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
python - <<PY
import csv, glob, collections, re
rows = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_conv3x3(?:_up1?)?<[^>]*>)", r["Kernel_Name"])
        if m:
            rows.setdefault((m.group(1).replace(" ", ""), r.get("Grid_Size", "")), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n# MFMA-busy share per conv layer (kernel instance, grid): MFMA busy SIMD-cycles / (kernel cycles x 1024 SIMDs); VALU-active share beside it")
for key, d in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        cyc = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"]) / 8.0
        mf = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
        va = sum(d["SQ_ACTIVE_INST_VALU"]) / len(d["SQ_ACTIVE_INST_VALU"]) * 4.0 if "SQ_ACTIVE_INST_VALU" in d else float("nan")
        print("%-40s grid %-10s  %7.1f us   MFMA busy %5.1f %%   VALU issue %5.1f %%" % (key[0], key[1], cyc / 2400.0, 100.0 * mf / (cyc * 1024.0), 100.0 * va / (cyc * 1024.0)))
PY
