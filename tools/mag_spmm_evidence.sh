#!/bin/bash
# the MAG-shaped aggregation (K = 128, mean, N = 1 939 743, 42.2 M entries; X = 993 MB does not fit the
# 256 MiB Infinity Cache) before and after the community reorder pass, with HBM-side traffic from FETCH_SIZE / WRITE_SIZE passes
# (separate rocprofv3 --pmc runs, counters only, calibrated on the 256 MiB copies of the same pass).
#   bash tools/mag_spmm_evidence.sh <outdir>
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$(realpath -m ${1:-$R/gpurun_out/r03/mag}); mkdir -p $O
export EGNN_PMC_K=128 EGNN_PMC_REDUCE=mean EGNN_PMC_GRAPHS=mag,mag_reordered EGNN_PMC_REPS=5
cd $R
timeout 900 python tools/spmm_pmc_workload.py > $O/timing.log 2>&1; echo "timing rc=$?"; grep us_per_call $O/timing.log
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/magpmc_$c; rm -rf $d
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python $R/tools/spmm_pmc_workload.py > $O/pmc_$c.log 2>&1); echo "pmc $c rc=$?"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|spmm|copyBuffer|FillFunctor|fill" $f > $O/pmc_$c.csv
  rm -rf $d
done
python3 - $O <<'PY'
import csv, json, sys, os
o = sys.argv[1]
COPY = 256 << 20
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(o, f"pmc_{c}.csv")
    if not os.path.exists(path):
        continue
    disp = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != c:
            continue
        e = disp.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0, int(r["Grid_Size"])])
        e[1] += float(r["Counter_Value"])
    order = [disp[k] for k in sorted(disp)]
    copies = sorted((v for k, v, g in order if "copyBuffer" in k), reverse=True)[:3]
    unit = COPY / (sum(copies) / len(copies))
    # per graph the workload issues 1 warm-up call + REPS timed calls; a call = one spmm_blk dispatch (+ its combine)
    reps = int(os.environ.get("EGNN_PMC_REPS", "5"))
    groups, cur, nblk = [], [], 0
    for k, v, g in order:
        if "spmm" not in k:
            continue
        if "spmm_blk" in k or "short_rows" in k:
            if nblk == reps + 1:
                groups.append(cur); cur, nblk = [], 0
            nblk += 1
        if nblk > 1:                      # skip the warm-up call
            cur.append((k, v))
    groups.append(cur)
    for name, grp in zip(("mag", "mag_reordered"), groups):
        per = {}
        for k, v in grp:
            kn = "spmm_blk_kernel" if "spmm_blk" in k else ("spmm_combine_kernel" if "combine" in k else k[:40])
            per.setdefault(kn, []).append(v)
        res.setdefault(name, {})[c] = dict(bytes_per_unit=unit, bytes_per_call=sum(sum(v) / len(v) for v in per.values()) * unit,
                                          launches={k: len(v) for k, v in per.items()})
for name, r in res.items():
    if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
        r["hbm_bytes_per_call"] = r["FETCH_SIZE"]["bytes_per_call"] + r["WRITE_SIZE"]["bytes_per_call"]
for line in open(os.path.join(o, "timing.log")):
    if line.startswith("us_per_call"):
        t = line.split()
        res.setdefault(t[3], {}).update(us_per_call=float(t[1]), nnz=int(t[5]), algorithmic_bytes=int(t[7]))
for name, r in res.items():
    if "us_per_call" in r and "algorithmic_bytes" in r:
        r["algorithmic_GBs"] = r["algorithmic_bytes"] / r["us_per_call"] * 1e-3
        r["frac_of_8TBs"] = r["algorithmic_GBs"] / 8000.0
        if "hbm_bytes_per_call" in r:
            r["traffic_over_algorithmic"] = r["hbm_bytes_per_call"] / r["algorithmic_bytes"]
json.dump(res, open(os.path.join(o, "mag_spmm_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:2500])
PY
