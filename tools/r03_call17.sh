#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/evidence; mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q -rfE --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -1 $O/pytest_gpu.log
timeout 900 python tools/kernel_bench.py --out $O/kernel_bench2.jsonl > $O/kernel_bench2.log 2>&1; grep '"nce"' $O/kernel_bench2.jsonl
