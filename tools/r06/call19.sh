#!/bin/bash
set +e
mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/layer_lab.jsonl
for sh in layer128 layer_169k; do timeout 600 tools/lab/gemm3_lab --shape $sh --iters 10 --rounds 5 >> gpurun_out/r06/layer_lab.jsonl 2>> gpurun_out/r06/layer_lab.err; done
python3 - gpurun_out/r06/layer_lab.jsonl <<'PY' | tee gpurun_out/r06/layer_lab.txt
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
shape = None
for r in rows:
    if r["shape"] != shape:
        shape = r["shape"]; print(f"== {shape}  M={r['M']} N={r['N']} K={r['K']}")
    print(f"  {r['variant']:42s} {r['us_med']:8.1f} us (min {r['us_min']:8.1f})  {r['tf_med']:6.1f} TF  {100*r['frac_of_417']:5.1f} %   err {r['mean_err']:.2e}")
PY
tail -2 gpurun_out/r06/layer_lab.err
