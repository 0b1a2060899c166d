#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call8; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" STEPS=3 timeout 150 python tools/checks/lsp_benchflow.py 2>&1 | grep -E "^#|^step|Error|error" | cut -c1-300; }
{
run SW=ownsum
run SW=dot
} > $O/out.txt 2>&1
cat $O/out.txt; ls -la $O; head -c 3000 $O/graph.dot
