#!/bin/bash
# One GPU-box session of the SpMM lab: sweeps on both graphs, then PMC passes for chosen variants.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02/lab; mkdir -p $O
for g in chunglu local; do
  timeout 300 $R/tools/lab/spmm_lab $R/tools/lab/data/$g.bin 256 > $O/sweep_${g}_K256.jsonl 2> $O/sweep_${g}_K256.err; echo "sweep $g rc=$?"
done
timeout 200 $R/tools/lab/spmm_lab $R/tools/lab/data/chunglu.bin 128 --only R128 > $O/sweep_chunglu_K128.jsonl 2>&1
for spec in ${PMC_SPECS:-"chunglu:seg_r01 chunglu:blk_R128_f5 local:seg_r01 local:blk_R128_f5 local:lds_R512_f5"}; do
  g=${spec%%:*}; v=${spec##*:}
  bash $R/tools/lab/pmc.sh $g $v $O/pmc ${PMC_GROUPS}
done
du -sh $O
