#!/bin/bash
# r04 call 1: LSP divergence trace, eager / graph / switches, dropout 0 vs 0.5
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call1; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" STEPS=14 timeout 150 python tools/checks/lsp_trace.py 2>&1 | grep -E "^#|^step|Error|error" ; }
{
run MODEL=sage KERNEL=rbf DROPOUT=0.5 GRAPH=0
run MODEL=sage KERNEL=rbf DROPOUT=0.0 GRAPH=0
run MODEL=sage KERNEL=rbf DROPOUT=0.5 GRAPH=1
run MODEL=sage KERNEL=cosine DROPOUT=0.5 GRAPH=0
run MODEL=sage KERNEL=rbf DROPOUT=0.5 GRAPH=0 EGNN_LSP_FULL_ROWS=0
run MODEL=gcn KERNEL=rbf DROPOUT=0.5 GRAPH=0
} > $O/lsp_trace.txt 2>&1
tail -c 6000 $O/lsp_trace.txt
