#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call19; mkdir -p $O
cd $R
for sec in "gemm,nce" "spmm,nce"; do timeout 300 python tools/kernel_bench.py --only $sec --quick --out $O/kb.jsonl > /dev/null 2>&1; echo "only=$sec $(grep '"nce"' $O/kb.jsonl)"; done
cd /tmp && timeout 400 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/profh -o h -- python $R/tools/kernel_bench.py --only gemm,nce --quick --out $O/kb_h.jsonl > $O/kb_h.log 2>&1
grep '"nce"' $O/kb_h.jsonl
find /tmp/profh -name "*hip_api_stats*" -exec cp {} $O/hip_api_stats.csv \; ; head -12 $O/hip_api_stats.csv | cut -c1-200
