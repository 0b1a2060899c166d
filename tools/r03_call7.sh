#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call7; mkdir -p $O
cd $R
echo "== graphed epoch tests alone"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -rfE --tb=line -p no:cacheprovider -k "graphed_epoch_replays" > $O/g1.log 2>&1; echo "rc=$?"; tail -3 $O/g1.log | cut -c1-200
echo "== same with EGNN_TRAIN_ROWS=0"; EGNN_TRAIN_ROWS=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -rfE --tb=line -p no:cacheprovider -k "graphed_epoch_replays" > $O/g2.log 2>&1; echo "rc=$?"; tail -3 $O/g2.log | cut -c1-200
echo "== same with EGNN_NCE_DMA=0"; EGNN_NCE_DMA=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -rfE --tb=line -p no:cacheprovider -k "graphed_epoch_replays" > $O/g3.log 2>&1; echo "rc=$?"; tail -3 $O/g3.log | cut -c1-200
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -rfE --tb=short -p no:cacheprovider -k "dma_form or fitnet or ppi_aux or rows_twins or sharded_epoch_captured or criteria_match or gemm_all_layouts or linear" > $O/new.log 2>&1; echo "rc=$?"; tail -3 $O/new.log | cut -c1-200; grep -E "^(FAILED|ERROR)|^E  " $O/new.log | cut -c1-400 | head -20
for v in 0 1 0 1; do echo "== bench EGNN_GEMM_DMA=$v"; EGNN_GEMM_DMA=$v timeout 600 python bench.py --steps 20 --warmup 3 --cpu-epochs 0 --no-parity --no-local-roofline 2>&1 | grep "^{" | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['phases_ms'], d['last_losses'])"; done
