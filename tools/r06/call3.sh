#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
echo "== dropin tests"; timeout 1200 python -m pytest tests/test_dropin_reference_scripts.py tests/test_gpu_training_parity.py -m gpu -q -x -p no:cacheprovider -k "reference_train_loop or accel" > gpurun_out/r06/t3.log 2>&1; echo rc=$?; tail -25 gpurun_out/r06/t3.log | cut -c1-400
echo "== refloop"; timeout 600 python tools/r06/refloop.py 2>&1 | grep -v Warn | tail -3 | cut -c1-900
