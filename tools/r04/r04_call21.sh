#!/bin/bash
# final validation of round 4: the full GPU suite, the driver's bench command, the sharded SAGE / MAG runs
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call21; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rfE ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log | cut -c1-300
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench_driver_cmd.log | tail -1 > $O/bench_driver_cmd.json; python3 -c "
import json; d=json.load(open('$O/bench_driver_cmd.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], d['repeat_blocks_ms_per_step'], 'traffic', r['traffic'], 'frac', r['frac'], 'parity', d['parity']['ok'], d['parity']['trajectory_dropout']['max_rel_err'], 'ref', d['reference_loop'].get('epochs_per_s'))"
grep real $O/bench_driver_cmd.log
for spec in "arxiv_lpw:--gnn sage --training lpw --steps 60" "mag:--workload mag --steps 5"; do name=${spec%%:*}; extra=${spec#*:}
  timeout 600 python bench.py --force-sharded --warmup 3 --cpu-epochs 0 $extra 2>$O/sharded_$name.err | grep "^{" | tail -1 > $O/sharded_1rank_$name.json; python3 -c "
import json; d=json.load(open('$O/sharded_1rank_$name.json')); print('$name', d['value'], d['ms_per_step'], d['launch'][:40], d['last_losses'])" || tail -5 $O/sharded_$name.err; done
