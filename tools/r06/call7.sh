#!/bin/bash
# de-phasing sweep: gpu_ms_per_replay of the headline epoch for several offsets of the BatchNorm kernels' output tensors
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for D in 0 4352 33024 69888 135424 528640 1052928 0; do
EGNN_DEPHASE_BYTES=$D EGNN_DEPHASE_ALL=${ALL:-0} timeout 600 python bench.py --steps 40 --warmup 5 --cpu-epochs 0 --no-parity --reference-epochs 0 --repeat-blocks 6 --no-local-roofline --settle-seconds 2 --probe-epochs 1 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); t=j['timing']
print('D=$D', 'value', j['value'], 'gpu_ms', t['gpu_ms_per_replay'], 'blocks', j['repeat_blocks_ms_per_step'])"
done
