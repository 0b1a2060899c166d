#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call11; mkdir -p $O
cd $R
( time timeout 1700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log | cut -c1-300
( time timeout 400 python bench.py ) > $O/bench_default.log 2>&1; echo "bench rc=$?"; tail -4 $O/bench_default.log | cut -c1-1500
grep '^{' $O/bench_default.log | tail -1 > $O/bench_default.json
for cfg in "gcn nce" "sage lpw"; do set -- $cfg; echo "== force-sharded $1 $2"; timeout 300 python bench.py --gnn $1 --training $2 --force-sharded --steps 40 --warmup 3 --cpu-epochs 0 --no-parity 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','launch','last_losses')}))
except Exception as e: print('FAILED', l[:800])
"; done | tee $O/sharded.txt
