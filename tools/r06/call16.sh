#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06/evidence
echo "== A/B predicated gathers (a = shipped, b = -DEGNN_SPMM_PRED)"; bash tools/r06/ab.sh 2>&1 | grep -E "build|value|mfma"
cp tools/r06/libegnn_hip_a.so efficient-gnns_amd/lib/libegnn_hip.so
echo "== mag 4 ranks sliced, one device"; timeout 1200 python bench.py --gpus 4 --one-device --agg sliced --workload mag --steps 2 --warmup 1 > gpurun_out/r06/evidence/one_device_4ranks_sliced_mag_bench.log 2>&1; echo rc=$?; tail -5 gpurun_out/r06/evidence/one_device_4ranks_sliced_mag_bench.log | cut -c1-400
grep '^{' gpurun_out/r06/evidence/one_device_4ranks_sliced_mag_bench.log | tail -1 > gpurun_out/r06/evidence/one_device_4ranks_sliced_mag_bench.json
bash tools/r06/evidence_r06.sh traffic
