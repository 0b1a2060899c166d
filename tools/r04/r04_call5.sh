#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call5; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" timeout 150 python tools/checks/lsp_benchflow.py 2>&1 | grep -E "^#|^step|Error|error" | cut -c1-300; }
{
run SW=x STEPS=12
run SW=gc STEPS=12
} > $O/lsp_benchflow2.txt 2>&1
cat $O/lsp_benchflow2.txt
