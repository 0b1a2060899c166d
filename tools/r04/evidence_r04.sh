#!/bin/bash
# Round-4 evidence session (one gpurun call): smoke, the full GPU suite, the default bench line, rocprof kernel stats of the same command,
# the secondary configs, the sharded path on one rank, the aggregation's PMC traffic on this build, the gather footprint sweep.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/evidence; mkdir -p $O
cd $R
bash tools/evidence.sh r04 smoke tests
echo "== bench (default arguments)"; ( time timeout 1200 python bench.py ) > $O/bench.log 2>&1; echo "rc=$?"; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json; cut -c1-260 $O/bench_line.json; tail -4 $O/bench.log | grep real
echo "== bench, f32-input MFMA pipeline"; EGNN_GEMM_PIPE=f32 timeout 900 python bench.py --cpu-epochs 0 --no-local-roofline --reference-epochs 0 --parity-trajectory-steps 0 --steps 100 2>&1 | grep "^{" | tail -1 > $O/bench_line_f32pipe.json; cut -c1-200 $O/bench_line_f32pipe.json
bash tools/evidence.sh r04 rocprof
bash tools/epoch_list.sh gpurun_out/r04/evidence/epoch_gcn_nce > /dev/null 2>&1; head -1 $O/epoch_gcn_nce/epoch.txt
echo "== secondary configs"; for cfg in "sage lpw" "sage lpw --kernel cosine" "gcn gpw" "gcn gpw --kernel rbf" "gcn kd" "sage nce" "gcn supervised"; do set -- $cfg
  echo "-- $cfg"; timeout 600 python bench.py --gnn $1 --training $2 $3 $4 --steps 100 --warmup 3 --cpu-epochs 0 --no-local-roofline --reference-epochs 0 2>&1 | grep "^{" | tail -1 | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); p=d.get('parity') or {}; t=p.get('trajectory_dropout') or {}
    print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'repeat_blocks_ms_per_step':d.get('repeat_blocks_ms_per_step'),'last_losses':d['last_losses'],'parity_ok':p.get('ok'),'loss_aux':p.get('loss_aux'),'max_rel_err':p.get('max_rel_err'),'grads':(p.get('grads') or {}).get('worst_violation_of_bar'),'trajectory_dropout':{'ok':t.get('ok'),'max_rel_err':t.get('max_rel_err'),'rtol':t.get('rtol')},'roofline_edges':d.get('roofline_edges')}), d['config']['workload'][:120])
except Exception as e: print('FAILED', l[:300])
"; done > $O/config_benches.txt 2>&1; grep -c value $O/config_benches.txt
echo "== sharded path, one rank over RCCL"
for spec in "arxiv:" "arxiv_eager:--graph off" "arxiv_lpw:--gnn sage --training lpw" "arxiv_gpw:--training gpw" "mag:--workload mag --steps 5"; do name=${spec%%:*}; extra=${spec#*:}
  timeout 600 python bench.py --force-sharded --steps 60 --warmup 3 --cpu-epochs 0 $extra 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_$name.json; python3 -c "
import json; d=json.load(open('$O/sharded_1rank_$name.json')); print('$name', d['value'], d['ms_per_step'], d['launch'][:40], d['last_losses'])"; done
bash tools/evidence.sh r04 traffic
echo "== gather footprint sweep"; timeout 300 python tools/checks/gather_footprint_sweep.py > $O/gather_footprint.txt 2>&1; cat $O/gather_footprint.txt
du -sh $O
