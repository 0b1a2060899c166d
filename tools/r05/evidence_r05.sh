#!/bin/bash
# Round-5 evidence session (one gpurun call): smoke, the full GPU suite (incl. the several-ranks-on-one-GPU tests, report kept), the
# default bench line + the driver's command, rocprof kernel stats of the same command, one ordered eager epoch, the secondary configs,
# the sharded path on one rank (+ per-kernel comparison with the single-GPU epoch), the aggregation's PMC traffic on this build, the
# full-size multi-rank comparison and the --one-device bench lines.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/evidence; mkdir -p $O
cd $R
export EGNN_MULTIRANK_REPORT_DIR=$O
bash tools/evidence.sh r05 smoke tests
echo "== bench (default arguments)"; ( time timeout 1200 python bench.py ) > $O/bench.log 2>&1; echo "rc=$?"; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json; cut -c1-260 $O/bench_line.json; tail -4 $O/bench.log | grep real
echo "== bench, the driver's command"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep "^{" | tail -1 > $O/bench_line_driver_cmd.json; cut -c1-200 $O/bench_line_driver_cmd.json
echo "== bench, f32-input MFMA pipeline"; EGNN_GEMM_PIPE=f32 timeout 900 python bench.py --cpu-epochs 0 --no-local-roofline --reference-epochs 0 --parity-trajectory-steps 0 --steps 100 2>&1 | grep "^{" | tail -1 > $O/bench_line_f32pipe.json; cut -c1-200 $O/bench_line_f32pipe.json
bash tools/evidence.sh r05 rocprof
bash tools/epoch_list.sh gpurun_out/r05/evidence/epoch_gcn_nce > /dev/null 2>&1; head -1 $O/epoch_gcn_nce/epoch.txt
echo "== secondary configs"; for cfg in "sage lpw" "sage lpw --kernel cosine" "gcn gpw" "gcn gpw --kernel rbf" "gcn kd" "sage nce" "gcn supervised"; do set -- $cfg
  echo "-- $cfg"; timeout 600 python bench.py --gnn $1 --training $2 $3 $4 --steps 100 --warmup 3 --cpu-epochs 0 --no-local-roofline --reference-epochs 0 2>&1 | grep "^{" | tail -1 | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); p=d.get('parity') or {}; t=p.get('trajectory_dropout') or {}
    print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'repeat_blocks_ms_per_step':d.get('repeat_blocks_ms_per_step'),'last_losses':d['last_losses'],'parity_ok':p.get('ok'),'loss_aux':p.get('loss_aux'),'max_rel_err':p.get('max_rel_err'),'grads':(p.get('grads') or {}).get('worst_violation_of_bar'),'trajectory_dropout':{'ok':t.get('ok'),'max_rel_err':t.get('max_rel_err'),'rtol':t.get('rtol')},'roofline_gemm':(d.get('roofline_gemm') or {}).get('frac')}), d['config']['workload'][:120])
except Exception as e: print('FAILED', l[:300])
"; done > $O/config_benches.txt 2>&1; grep -c value $O/config_benches.txt
echo "== sharded path, one rank over RCCL"
for spec in "arxiv:" "arxiv_eager:--graph off" "arxiv_lpw:--gnn sage --training lpw" "arxiv_gpw:--training gpw" "mag:--workload mag --steps 5"; do name=${spec%%:*}; extra=${spec#*:}
  timeout 600 python bench.py --force-sharded --steps 60 --warmup 3 --cpu-epochs 0 $extra 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_$name.json; python3 -c "
import json; d=json.load(open('$O/sharded_1rank_$name.json')); print('$name', d['value'], d['ms_per_step'], d['launch'][:40], d['last_losses'])"; done
bash tools/epoch_compare.sh gpurun_out/r05/evidence/epoch_compare > $O/epoch_compare.log 2>&1; head -3 $O/epoch_compare/compare.txt
echo "== several ranks on ONE GPU: full-size comparison (world 2) and the --one-device bench lines"
timeout 900 python tools/checks/multirank_one_gpu.py --world 2 --cases full-gcn-nce-static-natural-ov1,full-sage-lpw-natural-ov1 --out $O/multirank_fullsize_w2.json > $O/multirank_fullsize_w2.log 2>&1; grep "multirank w=\|MULTIRANK" $O/multirank_fullsize_w2.log
for W in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2961$W bench.py --gpus $W --one-device --steps 5 --warmup 2 2>&1 | grep "^{" | tail -1 > $O/one_device_${W}ranks_bench.json; cut -c1-120 $O/one_device_${W}ranks_bench.json
done
bash tools/evidence.sh r05 traffic
du -sh $O
