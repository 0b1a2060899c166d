#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for k in 8 4 2 8 4 2; do EGNN_NCE_KSPLIT=$k timeout 300 python tools/lab/nce_time.py 2>&1 | grep "nce fwd" | sed "s/^/KSPLIT=$k /"; done
