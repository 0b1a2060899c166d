#!/bin/bash
# r04 call 2: reproduce the bench.py lpw last_losses anomaly
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call2; mkdir -p $O
cd $R
B="--gnn sage --training lpw --steps 10 --warmup 3 --cpu-epochs 0 --no-local-roofline"
run() { echo "== $*"; timeout 200 python bench.py $B "$@" 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','last_losses')}), d['config'].get('graph'))
except Exception as e: print('FAILED', l[:600])
"; }
{
run
run --no-parity
run --no-parity --graph off
run --no-parity --probe-epochs 0
} > $O/lpw_repro.txt 2>&1
cat $O/lpw_repro.txt
