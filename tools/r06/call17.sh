#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
echo "== pytest -m gpu (final tree)"; (time timeout 2400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider) > gpurun_out/r06/pytest_gpu_final.log 2>&1; echo rc=$?; grep -E "passed|failed" gpurun_out/r06/pytest_gpu_final.log | tail -2
echo "== bench driver cmd (final tree)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r06/bench_final.json; python -c "
import json; j=json.load(open('gpurun_out/r06/bench_final.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['timing']['gpu_ms_per_replay'], j['reference_loop']['fraction_of_package_loop'], j['parity']['ok'])"
