#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call8; mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 1700 python -m pytest tests -m gpu -q -rfE --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-300
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_gpu.log | cut -c1-600 | head -20
echo "== one eager epoch"; bash tools/epoch_kernels.sh > $O/epoch_kernels.log 2>&1; cp gpurun_out/epoch_kernels/last_epoch.txt $O/epoch_kernels.txt; head -1 $O/epoch_kernels.txt
