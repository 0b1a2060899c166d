#!/bin/bash
# Ordered kernel list of ONE steady-state eager epoch (train step + eval) of the headline config: what is launched, how long.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/epoch_kernels; mkdir -p $O
rm -rf /tmp/profe
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/profe -o ek -- python $R/bench.py --graph off --steps 4 --warmup 2 --cpu-epochs 0 --no-parity --probe-epochs 0 --no-local-roofline > $O/run.log 2>&1); echo "rc=$?"
tail -1 $O/run.log | cut -c1-200
f=$(find /tmp/profe -name "*kernel_trace.csv" | head -1)
python3 - "$f" $O/last_epoch.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# an epoch = from one nce_fwd_kernel to the next; take the last complete one
idx = [i for i, n in enumerate(names) if "nce_fwd_kernel" in n]
a, b = idx[-2], idx[-1]
ep = rows[a:b]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ep)
span = int(ep[-1]["End_Timestamp"]) - int(ep[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write(f"# one eager epoch: {len(ep)} kernels, busy {tot/1e3:.1f} us, span {span/1e3:.1f} us\n")
    for r in ep:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        f.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:9.1f}  {n[:150]}\n")
print(open(sys.argv[2]).read()[:200])
PY
