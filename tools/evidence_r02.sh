#!/bin/bash
# One GPU-box session that collects the round's evidence into gpurun_out/r02/evidence/ (copied to profiles/r02_* afterwards).
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02/evidence; mkdir -p $O
cd $R
echo "== smoke"; timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?"; tail -1 $O/smoke.log | cut -c1-300
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -rfE --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -1 $O/pytest_gpu.log
echo "== kernel bench"; timeout 900 python tools/kernel_bench.py --out $O/kernel_bench.jsonl > $O/kernel_bench.log 2>&1; echo "rc=$?"
echo "== gemm lab"; EGNN_GEMM_PIPE=f32 timeout 300 tools/lab/gemm_lab --iters 10 > $O/gemm_lab_f32.jsonl 2>&1; timeout 300 tools/lab/gemm_lab --iters 10 > $O/gemm_lab_split.jsonl 2>&1; echo "rc=$?"
echo "== bench (default arguments)"; timeout 1200 python bench.py > $O/bench.log 2>&1; echo "rc=$?"; tail -1 $O/bench.log > $O/bench_line.json; cut -c1-250 $O/bench_line.json
echo "== rocprof kernel stats of bench"; rm -rf /tmp/profev; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profev -o r02 -- python $R/bench.py --steps 10 --warmup 2 --cpu-epochs 0 > $O/rocprof_bench.log 2>&1); echo "rc=$?"
find /tmp/profev -name "*kernel_stats*" -exec cp {} $O/bench_kernel_stats.csv \; ; find /tmp/profev -name "*domain_stats*" -exec cp {} $O/bench_domain_stats.csv \;
echo "== one eager epoch"; bash tools/epoch_kernels.sh > $O/epoch_kernels.log 2>&1; cp gpurun_out/epoch_kernels/last_epoch.txt $O/epoch_kernels.txt; head -1 $O/epoch_kernels.txt
echo "== secondary configs"; bash tools/config_benches.sh > $O/config_benches.txt 2>&1; grep -c value $O/config_benches.txt
echo "== sharded path, one rank over RCCL"; timeout 600 python bench.py --force-sharded --steps 10 --warmup 3 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_arxiv.json; cut -c1-200 $O/sharded_1rank_arxiv.json
timeout 900 python bench.py --force-sharded --workload mag --steps 5 --warmup 2 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_mag.json; cut -c1-200 $O/sharded_1rank_mag.json
for m in gpw lpw; do timeout 600 python bench.py --force-sharded --training $m --steps 10 --warmup 3 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_arxiv_$m.json; cut -c1-200 $O/sharded_1rank_arxiv_$m.json; done
du -sh $O
