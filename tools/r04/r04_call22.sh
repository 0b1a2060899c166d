#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call22; mkdir -p $O
cd $R
{ for v in product torchkl; do VARIANT=$v timeout 200 python tools/checks/graph_nodes.py 2>&1 | grep -E "^#|^replay|Error|error|assert" | cut -c1-400; done
  VARIANT=product TRAINING=nce MODEL=gcn timeout 200 python tools/checks/graph_nodes.py 2>&1 | grep -E "^#|^replay|Error|error|assert" | cut -c1-400; } > $O/graph_nodes.txt 2>&1
cat $O/graph_nodes.txt
