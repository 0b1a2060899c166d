#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call12; mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log | cut -c1-300
