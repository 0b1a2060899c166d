#!/bin/bash
# sampled heads + fused tail: parity tests, A/B bench lines, epoch listing
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call11; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "picked_rows or bn_act_linear or fused_tail or fused_bn or reference_goldens or grad_tap or criteria_rows" > $O/pytest_new.log 2>&1; echo "rc=$?"; tail -15 $O/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_full.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_full.log
B="--steps 20 --warmup 5 --cpu-epochs 0 --no-local-roofline"
echo "== default"; timeout 600 python bench.py $B 2>&1 | grep "^{" | tail -1 > $O/bench_default.json; cut -c1-260 $O/bench_default.json
echo "== EGNN_FUSED_TAIL=0"; EGNN_FUSED_TAIL=0 timeout 600 python bench.py $B --no-parity 2>&1 | grep "^{" | tail -1 > $O/bench_nofusedtail.json; cut -c1-260 $O/bench_nofusedtail.json
echo "== EGNN_SAMPLED_HEADS=0"; EGNN_SAMPLED_HEADS=0 timeout 600 python bench.py $B --no-parity 2>&1 | grep "^{" | tail -1 > $O/bench_nosampledheads.json; cut -c1-260 $O/bench_nosampledheads.json
echo "== both off"; EGNN_FUSED_TAIL=0 EGNN_SAMPLED_HEADS=0 timeout 600 python bench.py $B --no-parity 2>&1 | grep "^{" | tail -1 > $O/bench_bothoff.json; cut -c1-260 $O/bench_bothoff.json
echo "== gpw"; timeout 600 python bench.py --training gpw $B 2>&1 | grep "^{" | tail -1 > $O/bench_gpw.json; cut -c1-260 $O/bench_gpw.json
echo "== epoch"; bash tools/epoch_kernels.sh > $O/epoch_kernels.log 2>&1; cp gpurun_out/epoch_kernels/last_epoch.txt $O/epoch_kernels.txt; head -1 $O/epoch_kernels.txt
