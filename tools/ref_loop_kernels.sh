#!/bin/bash
# Ordered kernel list of ONE steady-state epoch of the reference's unmodified loop (arxiv_pyg/gnn.py through dropin/).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/$1; shift; mkdir -p $O
rm -rf /tmp/profr
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/profr -o rk -- python $R/bench.py --steps 2 --warmup 2 --cpu-epochs 0 --no-parity --probe-epochs 0 --no-local-roofline --repeat-blocks 0 --reference-epochs 6 "$@" > $O/run_ref.log 2>&1); echo "rc=$?"
grep '^{' $O/run_ref.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(json.dumps(d.get('reference_loop')))"
python3 - $(find /tmp/profr -name "*kernel_trace.csv" | head -1) $O/ref_epoch.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "ce_kd_fwd_kernel" in n]
a, b = idx[-2], idx[-1]
ep = rows[a:b]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ep)
span = int(ep[-1]["End_Timestamp"]) - int(ep[0]["Start_Timestamp"])
agg = collections.OrderedDict()
with open(sys.argv[2], "w") as f:
    f.write(f"# one epoch of the reference's own loop through dropin/: {len(ep)} kernels, busy {tot/1e3:.1f} us, span {span/1e3:.1f} us\n")
    for r in ep:
        n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:120]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        f.write(f"{d:9.1f}  {n}\n")
        c = agg.setdefault(n[:60], [0, 0.0]); c[0] += 1; c[1] += d
    f.write("# ---- by kernel\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{t:9.1f} {c:4d}  {k}\n")
print(open(sys.argv[2]).read()[-3500:])
PY
