#!/bin/bash
# kernel timeline of the sharded path (one rank): where is the GPU idle?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof_gaps
rm -rf /tmp/profs
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/profs -o sh -- python $R/bench.py --steps 6 --warmup 3 --cpu-epochs 0 > $R/gpurun_out/prof_gaps/run2.log 2>&1); echo "rc=$?"
grep -E "metric" $R/gpurun_out/prof_gaps/run2.log | cut -c1-200
f=$(find /tmp/profs -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
# last ~40 % of the run = steady state
rows = rows[int(n * 0.6):]
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = 0; end = t0; gaps = []
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > end:
        gaps.append((s - end, rows[i - 1]["Kernel_Name"][:70] if i else "", r["Kernel_Name"][:70]))
    if e > end:
        busy += e - max(s, end); end = e
print(f"window {(t1 - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms, kernels {len(rows)}")
gaps.sort(reverse=True)
print("top gaps (us, after kernel -> before kernel):")
for g in gaps[:40]:
    print(f"{g[0] / 1e3:9.1f}  {g[1]}  ->  {g[2]}")
import collections
c = collections.Counter()
for g in gaps: c[g[2]] += g[0]
print("gap time by NEXT kernel:")
for k, v in c.most_common(15): print(f"{v / 1e3:9.1f} us  {k}")
PY
