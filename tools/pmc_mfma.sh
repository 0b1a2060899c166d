#!/bin/bash
# MFMA utilisation + effective clock of the GEMM / G-CRD kernels (counters only, own pass).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/pmc2; mkdir -p /tmp/pmc2 $R/gpurun_out/pmc
cd /tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc2 -o mfma -- python $R/tools/kernel_bench.py --only gemm,nce --quick > $R/gpurun_out/pmc/mfma.log 2>&1; echo "rc=$?"
cd $R
f=$(find /tmp/pmc2 -name "*counter_collection.csv" | head -1); cp $f gpurun_out/pmc/mfma_counter_collection.csv
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmc/mfma_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "gemm_kernel" in k or "nce_" in k or "Cijk" in k:
        name = k.split("(")[0][-60:]
        agg[(name, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[(name, r["Grid_Size"])]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), c in agg.items():
    us = sum(c["us"]) / len(c["us"])
    gui = sum(c.get("GRBM_GUI_ACTIVE", [0])) / max(1, len(c.get("GRBM_GUI_ACTIVE", [1])))
    mf = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(1, len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [1])))
    bz = sum(c.get("SQ_BUSY_CYCLES", [0])) / max(1, len(c.get("SQ_BUSY_CYCLES", [1])))
    print(f"{name:62s} grid {grid:>9s} {us:9.1f} us  clk {gui/us/1e3:5.2f} GHz  mfma_busy/gui {mf/max(gui,1):8.2f}  sq_busy/gui {bz/max(gui,1):7.2f}")
PY
