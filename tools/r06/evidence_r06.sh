#!/bin/bash
# Round-6 evidence session (one gpurun call): smoke, the full GPU suite (multi-rank report kept), the default bench line + the driver's
# command, rocprof kernel stats of the same command + the replay-gap analysis of its kernel trace, one ordered eager epoch, the secondary
# configs, the sharded path on one rank, the aggregation's PMC traffic on this build, the full-size multi-rank comparisons (halo and
# column-sliced, arxiv and MAG) and the --one-device bench lines in both exchange forms.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06/evidence; mkdir -p $O
cd $R
export EGNN_MULTIRANK_REPORT_DIR=$O
STAGES=${@:-base configs sharded multirank traffic}
for s in $STAGES; do case $s in
base)
bash tools/evidence.sh r06 smoke
echo "== pytest -m gpu (the driver's command)"; ( time timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_gpu.log | head -1
echo "== bench (default arguments)"; ( time timeout 1200 python bench.py ) > $O/bench.log 2>&1; echo "rc=$?"; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json; cut -c1-260 $O/bench_line.json; tail -4 $O/bench.log | grep real
echo "== bench, the driver's command"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep "^{" | tail -1 > $O/bench_line_driver_cmd.json; cut -c1-200 $O/bench_line_driver_cmd.json
echo "== bench, f32-input MFMA pipeline"; EGNN_GEMM_PIPE=f32 timeout 900 python bench.py --cpu-epochs 0 --no-local-roofline --reference-epochs 0 --parity-trajectory-steps 0 --steps 100 2>&1 | grep "^{" | tail -1 > $O/bench_line_f32pipe.json; cut -c1-200 $O/bench_line_f32pipe.json
echo "== rocprof kernel stats + trace of the driver's command"; rm -rf /tmp/profev; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profev -o r06 -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-epochs 0 --reference-epochs 0 --no-parity > $O/rocprof_bench.log 2>&1); echo "rc=$?"
find /tmp/profev -name "*kernel_stats*" -exec cp {} $O/bench_kernel_stats.csv \; ; find /tmp/profev -name "*domain_stats*" -exec cp {} $O/bench_domain_stats.csv \; ; head -6 $O/bench_kernel_stats.csv | cut -c1-200
f=$(find /tmp/profev -name "*kernel_trace.csv" | head -1); python tools/r06/replay_gaps.py $f $O/replay_gaps.txt; head -2 $O/replay_gaps.txt
bash tools/epoch_list.sh gpurun_out/r06/evidence/epoch_gcn_nce > /dev/null 2>&1; head -1 $O/epoch_gcn_nce/epoch.txt
;;
configs)
echo "== secondary configs"; for cfg in "sage lpw" "sage lpw --kernel cosine" "gcn gpw" "gcn gpw --kernel rbf" "gcn kd" "sage nce" "gcn supervised"; do set -- $cfg
  echo "-- $cfg"; timeout 600 python bench.py --gnn $1 --training $2 $3 $4 --steps 100 --warmup 3 --cpu-epochs 0 --no-local-roofline --reference-epochs 0 2>&1 | grep "^{" | tail -1 | python3 -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); p=d.get('parity') or {}; t=p.get('trajectory_dropout') or {}
    print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'gpu_ms_per_replay':(d.get('timing') or {}).get('gpu_ms_per_replay'),'repeat_blocks_ms_per_step':d.get('repeat_blocks_ms_per_step'),'last_losses':d['last_losses'],'parity_ok':p.get('ok'),'loss_aux':p.get('loss_aux'),'max_rel_err':p.get('max_rel_err'),'grads':(p.get('grads') or {}).get('worst_violation_of_bar'),'trajectory_dropout':{'ok':t.get('ok'),'max_rel_err':t.get('max_rel_err'),'rtol':t.get('rtol')},'roofline_gemm':(d.get('roofline_gemm') or {}).get('frac')}), d['config']['workload'][:120])
except Exception as e: print('FAILED', l[:300])
"; done > $O/config_benches.txt 2>&1; grep -c value $O/config_benches.txt
;;
sharded)
echo "== sharded path, one rank over RCCL"
for spec in "arxiv:" "arxiv_eager:--graph off" "arxiv_lpw:--gnn sage --training lpw" "arxiv_gpw:--training gpw" "mag:--workload mag --steps 5"; do name=${spec%%:*}; extra=${spec#*:}
  timeout 600 python bench.py --force-sharded --steps 60 --warmup 3 --cpu-epochs 0 $extra 2>&1 | grep "^{" | tail -1 > $O/sharded_1rank_$name.json; python3 -c "
import json; d=json.load(open('$O/sharded_1rank_$name.json')); print('$name', d['value'], d['ms_per_step'], d['launch'][:40], d['last_losses'])"; done
;;
multirank)
echo "== several ranks on ONE GPU: full-size comparisons (world 2: halo / sliced; MAG at full size) and the --one-device bench lines"
timeout 1500 python tools/checks/multirank_one_gpu.py --world 2 --cases full-gcn-nce-static-natural-ov1,full-sage-lpw-natural-ov1,full-mag-sage-kd-ov1 --out $O/multirank_fullsize_w2.json > $O/multirank_fullsize_w2.log 2>&1; grep "multirank w=\|MULTIRANK" $O/multirank_fullsize_w2.log
timeout 1500 python tools/checks/multirank_one_gpu.py --world 4 --cases full-gcn-nce-static-sliced-natural-ov1,full-mag-sage-kd-sliced-ov1 --out $O/multirank_fullsize_sliced_w4.json > $O/multirank_fullsize_sliced_w4.log 2>&1; grep "multirank w=\|MULTIRANK" $O/multirank_fullsize_sliced_w4.log
for W in 2 4; do for A in halo sliced; do
  timeout 600 python bench.py --gpus $W --one-device --agg $A --steps 5 --warmup 2 2>/dev/null | grep "^{" | tail -1 > $O/one_device_${W}ranks_${A}_bench.json; python3 -c "
import json; d=json.load(open('$O/one_device_${W}ranks_${A}_bench.json')); c=d['comm_per_epoch']['per_rank'][0]; print('$W ranks $A', d['n_gpus'], d['value'], {k:v for k,v in c.items() if 'bytes' in k or 'exchanges' in k})"
done; done
# (the MAG-shaped graph at 4 ranks runs through tools/checks/multirank_one_gpu.py above: 37 s; bench.py --one-device --workload mag with 4 ranks spends
# > 20 min in its set-up on the box's 16 host threads -- locality pass + plans of a 42 M-entry graph in four processes -- and is not part of this session)
;;
traffic)
bash tools/evidence.sh r06 traffic
;;
esac; done
du -sh $O
