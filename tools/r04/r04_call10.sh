#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call10; mkdir -p $O
cd $R
timeout 200 python tools/checks/sharded_graph_check.py --gnn gcn --mode nce > $O/sharded_nce.log 2>&1; echo "rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/sharded_nce.log | tail -25 | cut -c1-1500
timeout 900 python -m pytest tests/test_gpu_training_parity.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_new.log 2>&1; echo "rc=$?"; tail -30 $O/pytest_new.log | cut -c1-400
