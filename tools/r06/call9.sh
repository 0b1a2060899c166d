#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
echo "== pytest -m gpu"; (time timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider) > gpurun_out/r06/pytest_gpu.log 2>&1; echo rc=$?; tail -12 gpurun_out/r06/pytest_gpu.log | cut -c1-300
