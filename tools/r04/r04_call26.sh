#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call26; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_training_parity.py tests/test_dropin_reference_scripts.py -m gpu -q --tb=short -p no:cacheprovider -k "accel or reference_train_loop" > $O/pytest_sel.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_sel.log | cut -c1-300
timeout 300 python bench.py --steps 60 --warmup 5 --cpu-epochs 0 --no-parity --no-local-roofline --probe-epochs 2 --repeat-blocks 0 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','reference_loop')}))" | tee $O/bench_ref.txt
