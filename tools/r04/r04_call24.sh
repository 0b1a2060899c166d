#!/bin/bash
# last full GPU suite of round 4 on the final tree + the driver's bench command
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call24; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rfE ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-300
