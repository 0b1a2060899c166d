#!/bin/bash
# One GPU-box session of the SpMM lab: sweeps on both graphs, then PMC passes for chosen variants.
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02/lab${LAB_TAG}; mkdir -p $O
for spec in ${SWEEPS:-chunglu:256 local:256 chunglu:128 chunglu:40 local:40}; do
  g=${spec%%:*}; k=${spec##*:}
  timeout 300 $R/tools/lab/spmm_lab $R/tools/lab/data/$g.bin $k > $O/sweep_${g}_K$k.jsonl 2> $O/sweep_${g}_K$k.err; echo "sweep $g K=$k rc=$?"
done
PMC_SPECS=${PMC_SPECS:-chunglu:seg_r01 chunglu:blk_R32_f0 local:blk_R32_f0 local:lds_R512_f0}
for spec in $PMC_SPECS; do
  g=${spec%%:*}; v=${spec##*:}
  bash $R/tools/lab/pmc.sh $g $v $O/pmc ${PMC_GROUPS:-sq_time tcp1 tcc1 tcc2 fetch write}
done
du -sh $O
