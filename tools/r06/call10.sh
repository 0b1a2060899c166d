#!/bin/bash
# scheduling-hint lab + the multirank suite with the column-sliced cases
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for sh in nce_bwd layer sq4k nce_fwd; do
  timeout 600 tools/lab/gemm3_lab --shape $sh --iters 5 --rounds 3 >> gpurun_out/r06/sched_lab.jsonl 2>> gpurun_out/r06/sched_lab.err
done
python3 - gpurun_out/r06/sched_lab.jsonl <<'PY' | tee gpurun_out/r06/sched_lab.txt
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
shape = None
for r in rows:
    if r["shape"] != shape:
        shape = r["shape"]; print(f"== {shape}  M={r['M']} N={r['N']} K={r['K']} splits={r['splits']}")
    print(f"  {r['variant']:42s} {r['us_med']:8.1f} us (min {r['us_min']:8.1f})  {r['tf_med']:6.1f} TF  {100*r['frac_of_417']:5.1f} %   err mean {r['mean_err']:.2e} max {r['max_err']:.2e}")
PY
tail -3 gpurun_out/r06/sched_lab.err
echo "== multirank"; timeout 2400 python -m pytest tests/test_gpu_multirank.py -q -m gpu -p no:cacheprovider > gpurun_out/r06/t10.log 2>&1; echo rc=$?; tail -6 gpurun_out/r06/t10.log | cut -c1-300
