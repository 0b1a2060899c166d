#!/bin/bash
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03/call20; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or skinny or linear or conv or nce or sage or bn_act or fused_tail or matmul" 2>&1 | tail -4
for sw in 1 0; do EGNN_SKINNY_TILE=$sw timeout 300 python tools/kernel_bench.py --only gemm,nce --quick --out $O/kb_$sw.jsonl > /dev/null 2>&1; echo "SKINNY_TILE=$sw $(grep 'xW3' $O/kb_$sw.jsonl | cut -c1-160) | $(grep '"nce"' $O/kb_$sw.jsonl | cut -c1-140)"; done
B="--steps 10 --warmup 3 --cpu-epochs 0 --no-local-roofline --no-parity"
for sw in 1 0; do echo "-- sage nce EGNN_SKINNY_TILE=$sw"; EGNN_SKINNY_TILE=$sw timeout 600 python bench.py --gnn sage --training nce $B 2>&1 | grep "^{" | tail -1 | cut -c90-200
echo "-- gcn nce EGNN_SKINNY_TILE=$sw"; EGNN_SKINNY_TILE=$sw timeout 600 python bench.py $B 2>&1 | grep "^{" | tail -1 | cut -c90-200; done
