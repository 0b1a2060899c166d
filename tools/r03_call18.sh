#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call18; mkdir -p $O
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profn -o n -- python $R/tools/kernel_bench.py --only nce --quick --out $O/kb_nce.jsonl > $O/kb.log 2>&1
cat $O/kb_nce.jsonl
find /tmp/profn -name "*kernel_stats*" -exec cp {} $O/nce_kernel_stats.csv \; ; head -14 $O/nce_kernel_stats.csv | cut -c1-100,180-300
cd $R; for k in 8 4 2; do EGNN_NCE_KSPLIT=$k timeout 200 python tools/kernel_bench.py --only nce --quick --out $O/kb_$k.jsonl > /dev/null 2>&1; echo "KSPLIT=$k $(grep nce $O/kb_$k.jsonl)"; done
EGNN_NCE_DMA=0 timeout 200 python tools/kernel_bench.py --only nce --quick --out $O/kb_nodma.jsonl > /dev/null 2>&1; echo "NCE_DMA=0 $(grep nce $O/kb_nodma.jsonl)"
