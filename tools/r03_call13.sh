#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call13; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "picked_rows or bn_act_linear or fused_tail or fused_bn" > $O/pytest_new.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_new.log
for v in 0 13; do EGNN_TAIL_ONE_PASS=0 EGNN_DXBN=$v timeout 300 python tools/lab/tail_time.py 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $O/tail_time.txt
cd /tmp && EGNN_TAIL_ONE_PASS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -o t -- python $R/tools/lab/tail_time.py > /dev/null 2>&1; find /tmp/proft -name "*kernel_stats*" -exec cp {} $O/tail_kernel_stats.csv \; ; grep -E "tail_bwd|apply_kernel<true, 3>|skinny_dx_kernel|rows_add|bwd_reduce" $O/tail_kernel_stats.csv | cut -c1-60,150-260
