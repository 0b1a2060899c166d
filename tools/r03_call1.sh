#!/bin/bash
# Round-3 GPU session 1: the tightened parity bars, the new full-size tests, the bench line with the driver-observed gather ceiling.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call1; mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 1700 python -m pytest tests -m gpu -q -rfE --tb=short -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | cut -c1-250 | head -40
echo "== bench (default)"; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log > $O/bench_line.json; cut -c1-1500 $O/bench_line.json
for cfg in "sage lpw" "sage lpw --kernel cosine" "gcn gpw" "gcn gpw --kernel rbf"; do set -- $cfg
  echo "== bench $cfg"; timeout 600 python bench.py --gnn $1 --training $2 $3 $4 --steps 5 --warmup 2 --cpu-epochs 0 --no-local-roofline > $O/bench_$1_$2_$4.log 2>&1; echo "rc=$?"
  tail -1 $O/bench_$1_$2_$4.log | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps({'value':d['value'],'parity':d['parity']})[:1200])
except Exception as e: print('FAILED', l[:600])
"; done
du -sh $O
