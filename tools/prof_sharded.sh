#!/bin/bash
# rocprofv3 kernel stats of the node-range sharded code path with one rank (the path the N>1 bench takes)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/prof_sharded
timeout 600 python -m pytest tests -m gpu -q -x -k "sharded_path" -p no:cacheprovider 2>&1 | tail -3
rm -rf /tmp/profs
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -o sh -- python $R/bench.py --force-sharded --steps 8 --warmup 2 --cpu-epochs 0 > $R/gpurun_out/prof_sharded/run.log 2>&1); echo "rc=$?"
grep -E "metric" $R/gpurun_out/prof_sharded/run.log | cut -c1-260
find /tmp/profs -name "*kernel_stats*" -exec cp {} $R/gpurun_out/prof_sharded/ \;
f=$(ls $R/gpurun_out/prof_sharded/*kernel_stats*.csv | head -1); head -40 $f | cut -c1-200
