#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call15; mkdir -p $O
cd $R
timeout 1700 python -m pytest tests -m gpu -q -rfE --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_gpu.log
B="--steps 20 --warmup 5 --cpu-epochs 0 --no-local-roofline"
echo "== default"; timeout 600 python bench.py $B 2>&1 | grep "^{" | tail -1 > $O/bench_default.json; cut -c1-260 $O/bench_default.json
echo "== EGNN_FUSED_TAIL=0"; EGNN_FUSED_TAIL=0 timeout 600 python bench.py $B --no-parity 2>&1 | grep "^{" | tail -1 > $O/bench_nofusedtail.json; cut -c1-260 $O/bench_nofusedtail.json
echo "== both off"; EGNN_FUSED_TAIL=0 EGNN_SAMPLED_HEADS=0 timeout 600 python bench.py $B --no-parity 2>&1 | grep "^{" | tail -1 > $O/bench_bothoff.json; cut -c1-260 $O/bench_bothoff.json
echo "== epoch"; bash tools/epoch_kernels.sh > $O/epoch_kernels.log 2>&1; cp gpurun_out/epoch_kernels/last_epoch.txt $O/epoch_kernels.txt; head -1 $O/epoch_kernels.txt
