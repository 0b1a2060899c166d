#!/bin/bash
# the driver's bench command on the final tree (after the last bench.py edits)
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/call28; mkdir -p $O
cd $R
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench_driver_cmd.log | tail -1 > $O/bench_driver_cmd.json; python3 -c "
import json; d=json.load(open('$O/bench_driver_cmd.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], len(d['repeat_blocks_ms_per_step']), d['repeat_blocks_median_ms_per_step'], 'traffic', r['traffic'], 'frac', r['frac'], 'parity', d['parity']['ok'], d['parity']['trajectory_dropout']['max_rel_err'], 'ref', d['reference_loop'].get('epochs_per_s'), d['reference_loop'].get('fraction_of_package_loop'), 'mfma', d['roofline_mfma']['frac'], 'cpu', d['cpu_baseline']['value'])"
grep real $O/bench_driver_cmd.log
