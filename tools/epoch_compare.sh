#!/bin/bash
# Per-kernel busy time of ONE steady-state eager epoch: single-GPU path vs the node-range sharded path on one rank.
# usage: tools/epoch_compare.sh <outdir> [bench args...]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/$1; shift; mkdir -p $O
for mode in single sharded; do
  extra=""; [ $mode = sharded ] && extra="--force-sharded"
  rm -rf /tmp/profc_$mode
  (cd /tmp && EGNN_BENCH_NORMAL_EXIT=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/profc_$mode -o ek -- python $R/bench.py --graph off --steps 4 --warmup 2 --cpu-epochs 0 --no-parity --probe-epochs 0 --no-local-roofline $extra "$@" > $O/run_$mode.log 2>&1); echo "$mode rc=$?"
  cp $(find /tmp/profc_$mode -name "*kernel_trace.csv" | head -1) $O/trace_$mode.csv
done
python3 - $O <<'PY'
import csv, sys, re, collections
O = sys.argv[1]
def epoch(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    mark = "ce_kd_fwd_kernel"
    idx = [i for i, n in enumerate(names) if mark in n]
    a, b = idx[-2], idx[-1]
    agg = collections.OrderedDict()
    for r in rows[a:b]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"void ", "", n)[:110]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        c = agg.setdefault(n, [0, 0.0]); c[0] += 1; c[1] += d
    return agg
s, h = epoch(f"{O}/trace_single.csv"), epoch(f"{O}/trace_sharded.csv")
keys = list(dict.fromkeys(list(s) + list(h)))
rows = [(k, *s.get(k, [0, 0.0]), *h.get(k, [0, 0.0])) for k in keys]
rows.sort(key=lambda r: -(r[4] - r[2]))
with open(f"{O}/compare.txt", "w") as f:
    f.write(f"# one eager epoch, busy us: single {sum(v[1] for v in s.values()):.1f} ({sum(v[0] for v in s.values())} launches)  sharded-1-rank {sum(v[1] for v in h.values()):.1f} ({sum(v[0] for v in h.values())} launches)\n")
    f.write("#  delta_us   single(n, us)    sharded(n, us)   kernel\n")
    for k, sn, su, hn, hu in rows:
        f.write(f"{hu - su:9.1f}   {sn:3d} {su:8.1f}    {hn:3d} {hu:8.1f}   {k}\n")
print(open(f"{O}/compare.txt").read()[:6000])
PY
