#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call20; mkdir -p $O
cd $R
for i in 1 2 3; do timeout 200 python tools/checks/sharded_graph_check.py --gnn gcn --mode gpw --static --port $((29600+i)) > $O/gpw_static_$i.log 2>&1; echo "gpw static run $i rc=$?"; done
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/gpw_static_1.log | tail -30 | cut -c1-400
timeout 600 python bench.py --force-sharded --workload mag --steps 5 --warmup 3 --cpu-epochs 0 > $O/mag.log 2>&1; echo "mag rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/mag.log | tail -25 | cut -c1-600
