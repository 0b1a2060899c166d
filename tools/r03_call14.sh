#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call14; mkdir -p $O
cd $R
for v in 16 32 0; do EGNN_TAILFWD=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "bn_act_linear or fused_tail" 2>&1 | tail -1; done
for v in 16 32 0; do EGNN_TAILFWD=$v timeout 300 python tools/lab/tail_time.py 2>&1 | grep "bn_act_linear" | sed "s/^/TAILFWD=$v /"; done | tee $O/tail_time.txt
EGNN_TAIL_ONE_PASS=0 timeout 300 python tools/lab/tail_time.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $O/tail_time.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -o t -- python $R/tools/lab/tail_time.py > /dev/null 2>&1; find /tmp/proft -name "*kernel_stats*" -exec cp {} $O/tail_kernel_stats.csv \; ; grep -E "tail_|skinny_fwd|bn_act_fwd" $O/tail_kernel_stats.csv | cut -c1-60,150-260
