#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, kernel microbench, bench line, rocprof kernel stats.
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log | cut -c1-300
echo "== kernel bench"; timeout 900 python tools/kernel_bench.py > gpurun_out/kernel_bench.log 2>&1; echo "kb rc=$?"; tail -40 gpurun_out/kernel_bench.log | cut -c1-400
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -5 gpurun_out/bench.log | cut -c1-3000
echo "== bench blas"; EGNN_GEMM=blas timeout 600 python bench.py --steps 20 --warmup 3 --cpu-epochs 0 > gpurun_out/bench_blas.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/bench_blas.log | cut -c1-3000
echo "== rocprof"; cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-epochs 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"; cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -25 $f | cut -c1-200; done
# keep the merge-back small: drop the raw trace, keep stats
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
