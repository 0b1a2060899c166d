#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call7; mkdir -p $O
cd $R
{ SYNC=1 timeout 120 python tools/probes/graph_reduce_repro.py; SYNC=0 timeout 120 python tools/probes/graph_reduce_repro.py; } > $O/graph_reduce_repro.txt 2>&1
cat $O/graph_reduce_repro.txt
