#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call22; mkdir -p $O
cd $R
timeout 170 python -m pytest tests/test_gpu_full_size.py tests/test_dropin_reference_scripts.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "lpw or lsp" > $O/pytest_lpw.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest_lpw.log
B="--steps 10 --warmup 3 --cpu-epochs 0 --no-local-roofline --no-parity"
for sw in 1 0; do echo "-- sage lpw EGNN_LSP_FULL_ROWS=$sw $(EGNN_LSP_FULL_ROWS=$sw timeout 100 python bench.py --gnn sage --training lpw $B 2>&1 | grep '^{' | tail -1 | cut -c90-200)"; done | tee $O/lpw_ab.txt
