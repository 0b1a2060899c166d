#!/bin/bash
# Round-3 GPU session 3: full-size parity against the float64 oracle, gemm3 ablations, the default bench line.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call3; mkdir -p $O
cd $R
echo "== gemm3 lab (ablations)"; LAB_TAG=_b LAB_ARGS="--shape sq4k" bash tools/lab/gemm3_round.sh | grep -E "shipped|abl_|pln_pln_128x128_bk16_nb3_p|f32k_pln_128x128_bk32_nb2_p|pln_pln_256"
LAB_TAG=_c LAB_ARGS="--shape nce_fwd --only abl_" bash tools/lab/gemm3_round.sh | grep -E "abl_"
echo "== pytest full size"; timeout 1500 python -m pytest tests/test_gpu_full_size.py -m gpu -q -rfE --tb=short -p no:cacheprovider > $O/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_full.log | cut -c1-300
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_full.log | cut -c1-400 | head -30
echo "== bench (default)"; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?"; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json; python3 -c "
import json
d=json.load(open('$O/bench_line.json')); print(d['value'], json.dumps(d['parity'])[:1800]); print(json.dumps(d['roofline'])[:900])"
