#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call23; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_training_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "chain_of_kernel or locality_order or flagged" > $O/pytest_sel.log 2>&1; echo "rc=$?"; tail -6 $O/pytest_sel.log | cut -c1-400
