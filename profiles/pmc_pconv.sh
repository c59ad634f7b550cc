cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_pconv; mkdir -p $OUT
cat > /tmp/run_precise_once.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mpiflow_amd.model import MPIPredictor
from mpiflow_amd.model.precise import PrecisePredictor
dev = torch.device("cuda:0"); S, H, W = 64, 384, 1280
m = MPIPredictor(W, H, S).randomize_(0).eval().to(dev)
pp = PrecisePredictor(m, dtype=torch.float32, x3=int(os.environ.get("MPF_PMC_X3", "0")))
img, dsp = torch.rand(1, 3, H, W, device=dev), torch.rand(1, 1, H, W, device=dev)
pp(img, dsp); torch.cuda.synchronize()
PY
cd /tmp
F='--kernel-include-regex k_pconv'
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY $F --output-format csv -d $OUT/a -o b -- python /tmp/run_precise_once.py > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE TA_BUSY_avr SQ_WAVE_CYCLES $F --output-format csv -d $OUT/b -o b -- python /tmp/run_precise_once.py > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE $F --output-format csv -d $OUT/c -o b -- python /tmp/run_precise_once.py > $OUT/c.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, re
def kname(n):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
rows = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        rows.setdefault((kname(r["Kernel_Name"]), r["Grid_Size"]), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    g = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"]) / 8.0 if "GRBM_GUI_ACTIVE" in d else 0
    if g * len(d.get("GRBM_GUI_ACTIVE", [])) < 2e6: continue
    mf = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
    va = sum(d["SQ_ACTIVE_INST_VALU"]) / len(d["SQ_ACTIVE_INST_VALU"]) * 4
    w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
    print("%-30s grid %-10s n=%d %8.1f us  MFMA busy %5.1f %%  VALU issue %5.1f %%  per wave: VALU %.0f SALU %.0f VMEM %.0f MFMA %.0f" % (k[0], k[1], len(d["SQ_WAVES"]), g / 2400.0, 100 * mf / (g * 1024), 100 * va / (g * 1024),
          sum(d["SQ_INSTS_VALU"]) / len(d["SQ_INSTS_VALU"]) / w, sum(d["SQ_INSTS_SALU"]) / len(d["SQ_INSTS_SALU"]) / w, sum(d["SQ_INSTS_VMEM_RD"]) / len(d["SQ_INSTS_VMEM_RD"]) / w, sum(d["SQ_INSTS_MFMA"]) / len(d["SQ_INSTS_MFMA"]) / w), end="")
    av = lambda n: sum(d[n]) / len(d[n]) if d.get(n) else float("nan")
    wc = av("SQ_WAVE_CYCLES")
    print("  | wait any %4.1f %% of wave cycles, vmem-active %4.1f %%, TA busy %4.1f %% | LDS: %.0f insts/wave, idx active %4.1f %% of CU cycles, conflict %4.1f %% of that, wait-lds %4.1f %% of wave cycles"
          % (100 * av("SQ_WAIT_ANY") / wc, 100 * av("SQ_ACTIVE_INST_VMEM") / wc, av("TA_BUSY_avr") / (g * 8) * 100 if g else 0, av("SQ_INSTS_LDS") / w,
             100 * av("SQ_LDS_IDX_ACTIVE") / (g * 256) if g else 0, 100 * av("SQ_LDS_BANK_CONFLICT") / max(av("SQ_LDS_IDX_ACTIVE"), 1), 100 * av("SQ_WAIT_INST_LDS") / wc))
PY
