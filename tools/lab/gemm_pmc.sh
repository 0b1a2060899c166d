#!/bin/bash
# rocprofv3 PMC passes over gemm_lab (counters only with --kernel-trace; one small group per pass).
#   tools/lab/gemm_pmc.sh <shape-substring> <outdir> [groups...]       EGNN_GEMM_PIPE selects the pipeline
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=$1; OUT=$(realpath -m $2); shift 2
declare -A G
G[sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
G[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA"
G[mem]="SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_BUSY_CU_CYCLES"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G[fetch]="FETCH_SIZE WRITE_SIZE"
mkdir -p $OUT
P=${EGNN_GEMM_PIPE:-split}
for g in ${@:-sq lds mem}; do
  d=/tmp/gpmc_$$_$g; rm -rf $d
  (cd /tmp && timeout 120 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $d -o p -- $R/tools/lab/gemm_lab --only $SHAPE --iters 2 > $OUT/${SHAPE}_${P}_$g.log 2>&1)
  rc=$?
  f=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|gemm_kernel" $f > $OUT/${SHAPE}_${P}_$g.csv; fi
  echo "pmc $SHAPE $P $g rc=$rc rows=$(wc -l < $OUT/${SHAPE}_${P}_$g.csv 2>/dev/null)"
  rm -rf $d
done
python3 - $OUT $SHAPE $P <<'PY'
import csv, glob, sys, os, collections
out, shape, pipe = sys.argv[1:4]
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(os.path.join(out, f"{shape}_{pipe}_*.csv")):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
avg = {k: v / max(len(n[k]), 1) for k, v in tot.items()}
print({k: round(v) for k, v in sorted(avg.items())})
wc = avg.get("SQ_WAVE_CYCLES", 0) or 1
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
    if k in avg: print(f"{k}/WAVE_CYCLES = {avg[k] / wc:.3f}")
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CYCLES" in avg: print("MFMA_BUSY/BUSY_CYCLES =", avg["SQ_VALU_MFMA_BUSY_CYCLES"] / avg["SQ_BUSY_CYCLES"])
if "SQ_LDS_IDX_ACTIVE" in avg: print("LDS conflict/idx_active =", avg.get("SQ_LDS_BANK_CONFLICT", 0) / max(avg["SQ_LDS_IDX_ACTIVE"], 1))
PY
