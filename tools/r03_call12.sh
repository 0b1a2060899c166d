#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call12; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "picked_rows or bn_act_linear or fused_tail or fused_bn" > $O/pytest_new.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_new.log
for v in 13 23 22 42 12; do EGNN_DXBN=$v timeout 300 python tools/lab/tail_time.py 2>&1 | grep -v Warning; done | tee $O/tail_time.txt
EGNN_TAIL_ONE_PASS=0 timeout 300 python tools/lab/tail_time.py 2>&1 | grep bn_act_linear | tee -a $O/tail_time.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -o t -- python $R/tools/lab/tail_time.py > /dev/null 2>&1; find /tmp/proft -name "*kernel_stats*" -exec cp {} $O/tail_kernel_stats.csv \; ; head -25 $O/tail_kernel_stats.csv | cut -c1-160
