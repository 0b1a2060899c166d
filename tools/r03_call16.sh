#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call16; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q --tb=short -p no:cacheprovider -k "sage or conv_layers or config3 or lpw or mag" 2>&1 | tail -3
B="--steps 10 --warmup 3 --cpu-epochs 0 --no-local-roofline"
for cfg in "sage lpw" "sage nce"; do set -- $cfg
  for sw in 1 0; do echo "-- $cfg EGNN_SAGE_NARROW_FIRST=$sw"; EGNN_SAGE_NARROW_FIRST=$sw timeout 600 python bench.py --gnn $1 --training $2 $B 2>&1 | grep "^{" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d.get('parity') or {}
print(d['value'], d['ms_per_step'], d['phases_ms'], 'parity', p.get('ok'), p.get('max_rel_err'))"; done; done | tee $O/sage.txt
