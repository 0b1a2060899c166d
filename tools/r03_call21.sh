#!/bin/bash
# per-kernel time split of the two secondary configs with their own kernels (LSP, GSP): input for the next round
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call21; mkdir -p $O
for cfg in "sage lpw" "gcn gpw"; do set -- $cfg
  rm -rf /tmp/profc; (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -o c -- python $R/bench.py --gnn $1 --training $2 --graph off --steps 8 --warmup 2 --cpu-epochs 0 --no-parity --probe-epochs 0 --no-local-roofline > $O/$1_$2.log 2>&1)
  find /tmp/profc -name "*kernel_stats*" -exec cp {} $O/$1_$2_kernel_stats.csv \;
  echo "== $cfg"; head -9 $O/$1_$2_kernel_stats.csv | cut -c1-70,120-230
done
