#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
echo "== sharded tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -p no:cacheprovider -k "sharded" > gpurun_out/r06/t18.log 2>&1; echo rc=$?; tail -4 gpurun_out/r06/t18.log | cut -c1-300
for spec in "arxiv:" "arxiv_lpw:--gnn sage --training lpw" "arxiv_gpw:--training gpw" "mag:--workload mag --steps 5"; do name=${spec%%:*}; extra=${spec#*:}
  timeout 600 python bench.py --force-sharded --steps 60 --warmup 3 --cpu-epochs 0 $extra 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r06/sharded_1rank_$name.json; python3 -c "
import json; d=json.load(open('gpurun_out/r06/sharded_1rank_$name.json')); print('$name', d['value'], d['ms_per_step'], d['launch'][:150], d['last_losses'])"; done
echo "== single-GPU line of the same session"; timeout 600 python bench.py --steps 60 --warmup 3 --cpu-epochs 0 --no-parity --reference-epochs 0 --no-local-roofline 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('single', d['value'], d['ms_per_step'])"
