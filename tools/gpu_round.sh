#!/bin/bash
# One GPU-box session. Usage: bash tools/gpu_round.sh [stages...]   (default: all)
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
STAGES=${@:-smoke tests kbench bench rocprof}
for s in $STAGES; do case $s in
smoke) echo "== smoke"; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log | cut -c1-400;;
tests) echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -rfE --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -60 gpurun_out/pytest_gpu.log | cut -c1-400;;
kbench) echo "== kernel bench"; timeout 900 python tools/kernel_bench.py $KBENCH_ARGS > gpurun_out/kernel_bench.log 2>&1; echo "kb rc=$?"; grep -v Warn gpurun_out/kernel_bench.log | tail -45 | cut -c1-400;;
bench) echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 $BENCH_ARGS > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-3000;;
rocprof) echo "== rocprof"; rm -rf /tmp/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r02 -- python $R/bench.py --steps 10 --warmup 2 --cpu-epochs 0 > $R/gpurun_out/rocprof.log 2>&1); echo "rocprof rc=$?";
   mkdir -p gpurun_out/prof; find /tmp/prof -name "*stats*" -exec cp {} gpurun_out/prof/ \; ; ls gpurun_out/prof; for f in $(ls gpurun_out/prof/*kernel_stats*.csv 2>/dev/null | head -1); do head -45 $f | cut -c1-260; done;;
esac; done
du -sh gpurun_out
