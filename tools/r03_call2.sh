#!/bin/bash
# Round-3 GPU session 2: where the full-size gradient differences come from + the gemm3 lab.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call2; mkdir -p $O
cd $R
echo "== full-size operator probe"; timeout 600 python tools/checks/fullsize_ops_probe.py > $O/ops_probe.log 2>&1; echo "rc=$?"; grep -v -i warn $O/ops_probe.log | tail -30
echo "== gemm3 lab"; LAB_TAG=_a bash tools/lab/gemm3_round.sh
echo "== gradient parity probe"; timeout 900 python tools/checks/grad_parity_probe.py --configs gcn:kd,sage:lpw:cosine > $O/grad_probe.log 2>&1; echo "rc=$?"; grep -v -i warn $O/grad_probe.log | tail -60
