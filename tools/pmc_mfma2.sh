#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/pmc3; mkdir -p /tmp/pmc3 $R/gpurun_out/pmc
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc3 -o sq -- python $R/tools/kernel_bench.py --only gemm --quick > $R/gpurun_out/pmc/sq.log 2>&1; echo "rc=$?"; tail -3 $R/gpurun_out/pmc/sq.log
cd $R
f=$(find /tmp/pmc3 -name "*counter_collection.csv" | head -1); cp $f gpurun_out/pmc/sq_counter_collection.csv
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/pmc/sq_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "gemm_kernel" in k or "Cijk" in k:
        name = (k.split("(")[0])[-46:]
        agg[(name, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_WAIT_INST_LDS","SQ_ACTIVE_INST_LDS"]
for (name, grid), c in agg.items():
    v = {n: (sum(c[n])/len(c[n]) if c.get(n) else 0) for n in names}
    wc = max(v["SQ_WAVE_CYCLES"], 1)
    print(f"{name:48s} grid {grid:>8s} wave_cyc {wc:12.0f} wait_any {v['SQ_WAIT_ANY']/wc:5.2f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:5.2f} active {v['SQ_ACTIVE_INST_ANY']/wc:5.2f} lds_conf/idx {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):5.2f} lds_idx/wc {v['SQ_LDS_IDX_ACTIVE']/wc:5.3f} wait_lds {v['SQ_WAIT_INST_LDS']/wc:5.2f}")
PY
