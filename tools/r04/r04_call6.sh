#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call6; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" STEPS=4 timeout 150 python tools/checks/lsp_trace.py 2>&1 | grep -E "^#|^step|Error|error" | cut -c1-900; }
{
run VARIANT=orig GRAPH=1 SYNC=1
run VARIANT=scalars GRAPH=1 SYNC=1
run VARIANT=refs GRAPH=1 SYNC=1
} > $O/lsp_sync.txt 2>&1
cat $O/lsp_sync.txt
