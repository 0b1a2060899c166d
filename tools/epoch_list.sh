#!/bin/bash
# Ordered + aggregated kernel list of ONE steady-state eager epoch of a bench configuration.  usage: tools/epoch_list.sh <outdir> [bench args]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/$1; shift; mkdir -p $O
rm -rf /tmp/profl
(cd /tmp && EGNN_BENCH_NORMAL_EXIT=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/profl -o ek -- python $R/bench.py --graph off --steps 4 --warmup 2 --cpu-epochs 0 --no-parity --probe-epochs 0 --no-local-roofline --reference-epochs 0 "$@" > $O/run.log 2>&1); echo "rc=$?"
python3 - $(find /tmp/profl -name "*kernel_trace.csv" | head -1) $O/epoch.txt "$*" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "ce_kd_fwd_kernel" in n]
a, b = idx[-2], idx[-1]
ep = rows[a:b]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ep)
agg = collections.OrderedDict()
with open(sys.argv[2], "w") as f:
    f.write(f"# bench.py {sys.argv[3]}: one eager epoch, {len(ep)} kernels, busy {tot/1e3:.1f} us\n")
    for r in ep:
        n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:110]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        f.write(f"{d:9.1f}  {n}\n")
        c = agg.setdefault(n[:70], [0, 0.0]); c[0] += 1; c[1] += d
    f.write("# ---- by kernel\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{t:9.1f} {c:4d}  {k}\n")
txt = open(sys.argv[2]).read()
print(txt[:150]); print(txt[txt.index("# ---- by kernel"):][:3500])
PY
