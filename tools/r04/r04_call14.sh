#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call14; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "sharded or colsum or tail or bn_act_linear or fused_bn or trajectory" > $O/pytest_sel.log 2>&1; echo "rc=$?"; tail -6 $O/pytest_sel.log | cut -c1-300
bash tools/epoch_compare.sh gpurun_out/r04/call14 2>&1 | head -40
for extra in "" "--force-sharded"; do timeout 300 python bench.py --steps 100 --warmup 3 --cpu-epochs 0 --no-parity --no-local-roofline $extra 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','launch','last_losses')}))
except Exception as e: print('FAILED', l[:800])
"; done | tee $O/bench.txt
