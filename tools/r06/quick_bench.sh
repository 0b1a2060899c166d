#!/bin/bash
# quick A/B line: value, replay time, the G-CRD kernels and the per-shape GEMM table (no CPU legs)
set +e
export TMPDIR=/tmp
timeout 600 python bench.py --steps 40 --warmup 5 --cpu-epochs 0 --no-parity --reference-epochs 0 --repeat-blocks 6 --no-local-roofline --settle-seconds 2 --probe-epochs 5 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); t=j['timing']
print('value', j['value'], 'gpu_ms', t['gpu_ms_per_replay'], 'blocks', j['repeat_blocks_ms_per_step'])
print('mfma', j['roofline_mfma']['ms_per_step'], j['roofline_mfma']['frac'], 'gemm', j['roofline_gemm']['ms_per_step'], j['roofline_gemm']['frac'], 'spmm', j['roofline']['avg_launch_us'])
for k,v in j['roofline_gemm']['by_shape'].items(): print('   ', k, v['us_per_call'], v['frac'])
"
