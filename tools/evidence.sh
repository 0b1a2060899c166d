#!/bin/bash
# One GPU-box session that collects a round's evidence into gpurun_out/<tag>/evidence/ (copied to profiles/<tag>_* afterwards).
#   bash tools/evidence.sh r03 [stages...]          stages: smoke tests bench rocprof epoch configs sharded traffic kbench
# The traffic stage writes profiles-ready JSON stamped with the SHA-256 of the libegnn_hip.so it ran on (bench.py reports the
# traffic only on that very build).
set +e
export TMPDIR=/tmp
TAG=${1:-r03}; shift
STAGES=${@:-smoke tests bench rocprof epoch configs sharded traffic kbench}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG/evidence; mkdir -p $O
cd $R
for s in $STAGES; do case $s in
smoke) echo "== smoke"; timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?"; tail -1 $O/smoke.log | cut -c1-300;;
tests) echo "== pytest -m gpu"; timeout 1700 python -m pytest tests -m gpu -q -rfE --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -1 $O/pytest_gpu.log;;
bench) echo "== bench (default arguments)"; timeout 1200 python bench.py > $O/bench.log 2>&1; echo "rc=$?"; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json; cut -c1-300 $O/bench_line.json
       echo "== bench, f32-input MFMA pipeline"; EGNN_GEMM_PIPE=f32 timeout 900 python bench.py --cpu-epochs 0 --no-local-roofline 2>&1 | grep "^{" | tail -1 > $O/bench_line_f32pipe.json; cut -c1-200 $O/bench_line_f32pipe.json;;
rocprof) echo "== rocprof kernel stats of bench"; rm -rf /tmp/profev; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profev -o $TAG -- python $R/bench.py --steps 10 --warmup 2 --cpu-epochs 0 > $O/rocprof_bench.log 2>&1); echo "rc=$?"
       find /tmp/profev -name "*kernel_stats*" -exec cp {} $O/bench_kernel_stats.csv \; ; find /tmp/profev -name "*domain_stats*" -exec cp {} $O/bench_domain_stats.csv \; ; head -8 $O/bench_kernel_stats.csv | cut -c1-200;;
epoch) echo "== one eager epoch"; bash tools/epoch_kernels.sh > $O/epoch_kernels.log 2>&1; cp gpurun_out/epoch_kernels/last_epoch.txt $O/epoch_kernels.txt; head -1 $O/epoch_kernels.txt;;
configs) echo "== secondary configs"; for cfg in "sage lpw" "sage lpw --kernel cosine" "gcn gpw" "gcn gpw --kernel rbf" "gcn kd" "sage nce" "gcn supervised"; do set -- $cfg
         echo "-- $cfg"; timeout 600 python bench.py --gnn $1 --training $2 $3 $4 --steps 10 --warmup 3 --cpu-epochs 0 --no-local-roofline 2>&1 | grep "^{" | tail -1 | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); p=d.get('parity') or {}
    print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'phases_ms':d['phases_ms'],'last_losses':d['last_losses'],'parity_ok':p.get('ok'),'loss_aux':p.get('loss_aux'),'max_rel_err':p.get('max_rel_err'),'grads':(p.get('grads') or {}).get('worst_violation_of_bar')}), d['config']['workload'][:120])
except Exception as e: print('FAILED', l[:300])
"; done > $O/config_benches.txt 2>&1; grep -c value $O/config_benches.txt
         echo "-- S=8192"; timeout 600 python bench.py --steps 20 --warmup 3 --cpu-epochs 0 --max-samples 8192 --no-parity --no-local-roofline 2>&1 | grep "^{" | tail -1 | cut -c1-300 >> $O/config_benches.txt;;
sharded) echo "== sharded path, one rank over RCCL"; timeout 600 python bench.py --force-sharded --steps 15 --warmup 3 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_arxiv.json; cut -c1-200 $O/sharded_1rank_arxiv.json
         timeout 600 python bench.py --force-sharded --graph off --steps 15 --warmup 3 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_arxiv_eager.json; cut -c1-200 $O/sharded_1rank_arxiv_eager.json
         timeout 900 python bench.py --force-sharded --workload mag --steps 5 --warmup 2 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_mag.json; cut -c1-200 $O/sharded_1rank_mag.json
         for m in gpw lpw; do timeout 600 python bench.py --force-sharded --training $m --steps 10 --warmup 3 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_arxiv_$m.json; cut -c1-200 $O/sharded_1rank_arxiv_$m.json; done;;
traffic) echo "== aggregation HBM-side traffic (PMC passes on this build)"
         for g in chunglu local; do bash tools/lab/pmc.sh $g default $O/pmc fetch write; done
         python tools/lab/traffic.py $O/pmc chunglu default 367506008 > $O/spmm_traffic.json; python3 -c "import json; d=json.load(open('$O/spmm_traffic.json')); print(d['lib_sha16'], d['hbm_bytes_per_call'], d['traffic_over_algorithmic'])"
         python tools/lab/traffic.py $O/pmc local default 367506008 > $O/spmm_traffic_local.json; python3 -c "import json; d=json.load(open('$O/spmm_traffic_local.json')); print(d['lib_sha16'], d['hbm_bytes_per_call'], d['traffic_over_algorithmic'])";;
kbench) echo "== kernel bench"; timeout 900 python tools/kernel_bench.py --out $O/kernel_bench.jsonl > $O/kernel_bench.log 2>&1; echo "rc=$?"; tail -3 $O/kernel_bench.log | cut -c1-200;;
esac; done
du -sh $O
