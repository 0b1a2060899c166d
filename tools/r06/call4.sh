#!/bin/bash
set +e
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/r06
rm -rf /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r06 -- python $R/bench.py --steps 20 --warmup 5 --cpu-epochs 0 --no-parity --reference-epochs 0 --repeat-blocks 0 --no-local-roofline --settle-seconds 1 --probe-epochs 3 > $R/gpurun_out/r06/trace_bench.log 2>&1); echo rc=$?
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); echo $f; wc -l $f
python tools/r06/replay_gaps.py $f gpurun_out/r06/replay_gaps.txt; head -3 gpurun_out/r06/replay_gaps.txt
python - "$f" <<'PY'
import csv, sys, gzip
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]))
rows.sort()
with gzip.open("gpurun_out/r06/kernel_trace_min.csv.gz","wt") as g:
    for s,e,n in rows: g.write(f"{s},{e},{n}\n")
PY
ls -la gpurun_out/r06/
