#!/bin/bash
# replay-vs-eager per-kernel times, with and without the de-phasing lab switch; then the new tests
set +e
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/r06
for D in 0 4352 69888; do
rm -rf /tmp/prof
(cd /tmp && EGNN_DEPHASE_BYTES=$D timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r06 -- python $R/bench.py --steps 20 --warmup 5 --cpu-epochs 0 --no-parity --reference-epochs 0 --repeat-blocks 0 --no-local-roofline --settle-seconds 1 --probe-epochs 3 > $R/gpurun_out/r06/trace_bench_$D.log 2>&1); echo "D=$D rc=$?"
tail -1 $R/gpurun_out/r06/trace_bench_$D.log | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('value', j['value'], j['timing']['gpu_ms_per_replay'])"
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]))
rows.sort()
names=[r[2] for r in rows]
lastcopy=[i for i,n in enumerate(names) if 'copyBuffer' in n]
# replays: region between copyBuffers with ~112 kernels
agg=collections.defaultdict(list); cnt=0
for a,b in zip(lastcopy[:-1], lastcopy[1:]):
    if 105 <= b-a <= 118:
        cnt+=1
        for s,e,n in rows[a+1:b]: agg[n].append((e-s)/1e3)
last=lastcopy[-1]
eag=collections.defaultdict(list)
for s,e,n in rows[last+1:]: eag[n].append((e-s)/1e3)
print("replays", cnt, "eager kernels", len(rows)-last-1)
for key in ("bn_act_bwd_apply_kernel<true, 0>","tail_fwd_tile_kernel<3, true>","bn_act_fwd_kernel","bn_act_bwd_reduce_kernel","tail_bwd_tile","bn_act_bwd_apply_kernel<true, 3>","spmm_blk_kernel<false, true","nce_fwd"):
    r=[v for n,vs in agg.items() if key in n for v in vs]; e=[v for n,vs in eag.items() if key in n for v in vs]
    if r and e: print("%-40s replay %8.1f   eager %8.1f  (per epoch)"%(key, sum(r)/cnt, sum(e)/3))
print("total per epoch replay %.1f eager %.1f"%(sum(sum(v) for v in agg.values())/cnt, sum(sum(v) for v in eag.values())/3))
PY
done
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training_parity.py -m gpu -q -x -p no:cacheprovider -k "graphed or accuracy or trajectory" > gpurun_out/r06/t6.log 2>&1; echo rc=$?; tail -4 gpurun_out/r06/t6.log | cut -c1-300
