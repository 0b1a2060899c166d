#!/bin/bash
# round 6, first GPU session: new tests, driver's bench command, sysfs clock files
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
echo "== sysfs"; ls /sys/class/drm/ 2>&1 | head; for c in /sys/class/drm/card*/device; do echo $c; ls $c | grep -E "pp_dpm|hwmon|power" | head -20; cat $c/pp_dpm_sclk 2>&1 | head -5; done
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "graphed_epoch or constant_operand or cut_once or linear_rows" > gpurun_out/r06/t1.log 2>&1; echo rc=$?; tail -5 gpurun_out/r06/t1.log | cut -c1-300
echo "== bench (driver cmd)"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_driver.json 2> gpurun_out/r06/bench_driver.err; echo rc=$?
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/bench_driver.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','timing','repeat_blocks_ms_per_step','eager','phases_ms'): print(k, j.get(k))
print('roofline', {k:v for k,v in j['roofline'].items() if k in ('frac','avg_launch_us','mfma_frac','gemm_frac')})
print('ref_loop', {k:v for k,v in (j.get('reference_loop') or {}).items() if k in ('epochs_per_s','fraction_of_package_loop','error')})
print('parity ok', j['parity']['ok'], j['parity']['max_rel_err'], j['parity']['trajectory_dropout']['max_rel_err'])
print('cpu', j['cpu_baseline']['value'])
PY
tail -3 gpurun_out/r06/bench_driver.err | cut -c1-300
