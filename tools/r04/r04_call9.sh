#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call9; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_training_parity.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_new.log 2>&1; echo "rc=$?"; tail -30 $O/pytest_new.log
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q --tb=short -p no:cacheprovider -k "lpw or lsp or graphed or sharded or split_acc or bias" > $O/pytest_sel.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_sel.log
B="--steps 10 --warmup 3 --cpu-epochs 0 --no-local-roofline --no-parity"
for cfg in "sage lpw" "gcn nce"; do set -- $cfg; echo "== $1 $2"; timeout 200 python bench.py --gnn $1 --training $2 $B 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','last_losses')}))
except Exception as e: print('FAILED', l[:800])
"; done | tee $O/bench.txt
