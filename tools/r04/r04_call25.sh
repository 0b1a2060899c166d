#!/bin/bash
# L2-miss fabric traffic of the LSP edge kernels (separate --pmc passes with --kernel-trace only, as the guide prescribes)
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call25; mkdir -p $O
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_edges_$c; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o p -- python $R/bench.py --gnn sage --training lpw --graph off --steps 3 --warmup 1 --cpu-epochs 0 --no-parity --probe-epochs 0 --no-local-roofline --reference-epochs 0 > $O/run_$c.log 2>&1); echo "$c rc=$?"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c > $O/edges_$c.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if not any(s in k for s in ("edge_sim_kernel", "lsp_loss_fwd_kernel", "lsp_loss_bwd_kernel", "seg_sum_kernel", "edge_coef_kernel")):
        continue
    key = k.split("(")[0][-40:]
    c = agg.setdefault(key, [])
    c.append(float(r["Counter_Value"]))
print("#", sys.argv[2], "per launch (counter units), launches in dispatch order")
for k, v in agg.items():
    print(k, len(v), [round(x, 1) for x in v[:8]])
PY
  cat $O/edges_$c.txt
done
