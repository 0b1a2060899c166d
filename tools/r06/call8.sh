#!/bin/bash
# full GPU suite + smoke + the driver's bench command
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > gpurun_out/r06/smoke.log 2>&1; echo rc=$?; tail -1 gpurun_out/r06/smoke.log | cut -c1-300
echo "== pytest -m gpu"; (time timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider) > gpurun_out/r06/pytest_gpu.log 2>&1; echo rc=$?; tail -8 gpurun_out/r06/pytest_gpu.log | cut -c1-300
echo "== bench (driver cmd)"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_driver.json 2> gpurun_out/r06/bench_driver.err; echo rc=$?
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r06/bench_driver.json') if l.startswith('{"metric"')][-1])
for k in ('value','ms_per_step','repeat_blocks_ms_per_step'): print(k, j.get(k))
t=j['timing']; print({k:t[k] for k in ('gpu_ms_per_replay','host_gap_ms_per_step','ms_per_step_read_then_launch','device_clocks_under_load')}); print(t['settle_before_warmup']['block_ms_per_step'])
print('roofline', {k:v for k,v in j['roofline'].items() if k in ('frac','avg_launch_us','mfma_frac','gemm_frac')})
r=j.get('reference_loop') or {}; print('ref_loop', {k:(v if not isinstance(v,dict) else v.get('epochs_per_s')) for k,v in r.items() if k!='what'})
print('parity ok', j['parity']['ok'], j['parity']['max_rel_err'], j['parity']['trajectory_dropout']['max_rel_err'])
print('cpu', j['cpu_baseline']['value'])
PY
