#!/bin/bash
# rocprofv3 PMC passes over the stand-alone lab driver (counters only with --kernel-trace; one small group per pass).
#   tools/lab/pmc.sh <graph> <variant> <outdir> [groups...]
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
GRAPH=$1; VAR=$2; OUT=$(realpath -m $3); shift 3
GROUPS_ALL="sq_time sq_inst tcp1 tcp3 tcc1 tcc2 ta"
declare -A G
G[sq_time]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
G[sq_inst]="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES"
G[tcp1]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
G[tcp3]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
G[tcc1]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
G[tcc2]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
G[ta]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
mkdir -p $OUT
for g in ${@:-$GROUPS_ALL}; do
  d=/tmp/pmc_$$_$g; rm -rf $d
  (cd /tmp && timeout 120 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $d -o p -- $R/tools/lab/spmm_lab $R/tools/lab/data/$GRAPH.bin 256 --exact $VAR --pmc > $OUT/${GRAPH}_${VAR}_$g.log 2>&1)
  rc=$?
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|spmm|bn_stats|copyBuffer" $f > $OUT/${GRAPH}_${VAR}_$g.csv; fi
  echo "pmc $GRAPH $VAR $g rc=$rc rows=$(wc -l < $OUT/${GRAPH}_${VAR}_$g.csv 2>/dev/null)"
  rm -rf $d
done
