#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call17; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_training_parity.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or sage or edge_sim or lsp or lpw or conv_layers" > $O/pytest_sel.log 2>&1; echo "rc=$?"; tail -8 $O/pytest_sel.log | cut -c1-300
B="--steps 60 --warmup 3 --cpu-epochs 0 --no-local-roofline --no-parity --reference-epochs 0"
for cfg in "sage lpw" "sage lpw --kernel cosine" "sage nce" "gcn nce"; do echo "== $cfg"; timeout 200 python bench.py --gnn $(echo $cfg | cut -d' ' -f1) --training $(echo $cfg | cut -d' ' -f2-) $B 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps({k:d.get(k) for k in ('value','ms_per_step','last_losses','repeat_blocks_ms_per_step')}))
except Exception as e: print('FAILED', l[:800])
"; done | tee $O/bench.txt
