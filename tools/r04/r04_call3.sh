#!/bin/bash
# r04 call 3: which replay goes wrong, and do held references change it
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call3; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" STEPS=8 timeout 150 python tools/checks/lsp_trace.py 2>&1 | grep -E "^#|^step|Error|error" | cut -c1-420; }
{
run VARIANT=orig GRAPH=1
run VARIANT=scalars GRAPH=1
run VARIANT=refs GRAPH=1
run VARIANT=orig GRAPH=1 DROPOUT=0.0
run VARIANT=orig GRAPH=1 MODEL=gcn
run VARIANT=orig GRAPH=0
} > $O/lsp_variants.txt 2>&1
cat $O/lsp_variants.txt
