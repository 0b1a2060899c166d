#!/bin/bash
# Round-3 GPU session 6: whole GPU suite, headline bench, sharded one-rank runs (captured vs eager), MAG aggregation evidence.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call6; mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 1700 python -m pytest tests -m gpu -q -rfE --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-300
grep -E "^(FAILED|ERROR)|^E  " $O/pytest_gpu.log | cut -c1-600 | head -20
echo "== bench (default)"; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?"; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json; python3 -c "
import json
d=json.load(open('$O/bench_line.json')); print(d['value'], d['ms_per_step'], d['phases_ms'], d['eager'], d['roofline_mfma']['frac'], d['parity']['ok'])"
for g in on off; do
  echo "== sharded one rank, nce, --graph $g"; timeout 600 python bench.py --force-sharded --graph $g --steps 15 --warmup 3 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_nce_$g.json
  python3 -c "
import json
d=json.load(open('$O/sharded_nce_$g.json')); print(d['value'], d['ms_per_step'], d.get('launch'), json.dumps(d.get('comm_per_epoch',{}).get('per_rank'))[:500])"
done
echo "== sharded one rank, sage lpw / kd, --graph on"; for m in lpw kd; do timeout 600 python bench.py --force-sharded --gnn sage --training $m --graph on --steps 10 --warmup 3 --cpu-epochs 0 2>&1 | grep "^{" | tail -1 > $O/sharded_$m.json; python3 -c "
import json
d=json.load(open('$O/sharded_$m.json')); print('$m', d['value'], d['ms_per_step'], d.get('launch','')[:80])"; done
echo "== MAG aggregation evidence"; bash tools/mag_spmm_evidence.sh $O/mag
