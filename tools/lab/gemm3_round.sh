#!/bin/bash
# One GPU-box session of the gemm3 lab (csrc/gemm3.h variants vs the shipped kernel, interleaved rounds in one process).
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/gemm3${LAB_TAG}; mkdir -p $O
timeout ${LAB_TIMEOUT:-900} $R/tools/lab/gemm3_lab $LAB_ARGS > $O/lab.jsonl 2> $O/lab.err; echo "gemm3_lab rc=$?"
python3 - $O/lab.jsonl <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
shape = None
for r in rows:
    if r["shape"] != shape:
        shape = r["shape"]; print(f"== {shape}  M={r['M']} N={r['N']} K={r['K']} splits={r['splits']}")
    print(f"  {r['variant']:34s} lds {r['lds']:6d}  {r['us_med']:8.1f} us (min {r['us_min']:8.1f})  {r['tf_med']:6.1f} TF  {100*r['frac_of_417']:5.1f} %   err mean {r['mean_err']:.2e} max {r['max_err']:.2e}")
PY
tail -3 $O/lab.err
