#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call27; mkdir -p $O
cd $R
for k in cosine rbf; do timeout 200 python bench.py --gnn gcn --training gpw --kernel $k --steps 40 --warmup 3 --cpu-epochs 0 --no-parity --no-local-roofline --reference-epochs 0 --repeat-blocks 0 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$k', json.dumps({k:d.get(k) for k in ('value','ms_per_step','roofline_gsp')}))"; done | tee $O/gsp_roofline.txt
