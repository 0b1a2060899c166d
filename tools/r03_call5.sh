#!/bin/bash
# Round-3 GPU session 5: G-CRD backward on the DMA pipeline (A/B), its parity tests, the full-size tests, bench.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call5; mkdir -p $O
cd $R
echo "== nce A/B"; for v in 0 1 0 1; do EGNN_NCE_DMA=$v timeout 300 python tools/lab/nce_time.py 2>&1 | grep "nce fwd" | sed "s/^/EGNN_NCE_DMA=$v /"; done
echo "== pytest nce + full size"; timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -rfE --tb=short -p no:cacheprovider -k "nce or full_size or lsp or segment_softmax or goldens or graphed or sharded" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
grep -E "^(FAILED|ERROR)|^E  " $O/pytest.log | cut -c1-1200 | head -20
echo "== bench (default)"; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "bench rc=$?"; grep "^{" $O/bench.log | tail -1 > $O/bench_line.json; python3 -c "
import json
d=json.load(open('$O/bench_line.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline_mfma'])[:600]); print(json.dumps(d['parity']['grads'])[:400], d['parity']['ok'])"
echo "== bench lpw"; timeout 600 python bench.py --gnn sage --training lpw --steps 5 --warmup 2 --cpu-epochs 0 --no-local-roofline > $O/bench_lpw.log 2>&1; echo "rc=$?"; grep "^{" $O/bench_lpw.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['parity']; print(d['value'], p['ok'], p['loss_aux'], p['max_rel_err'], p['losses_ok'])"
