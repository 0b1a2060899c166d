#!/bin/bash
# One GPU-box session of the GEMM lab: both pipelines on the epoch's shapes.
set +e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02/gemm${LAB_TAG}; mkdir -p $O
EGNN_GEMM_PIPE=f32 timeout 300 $R/tools/lab/gemm_lab $LAB_ARGS > $O/f32.jsonl 2> $O/f32.err; echo "f32 rc=$?"
timeout 300 $R/tools/lab/gemm_lab $LAB_ARGS > $O/split.jsonl 2> $O/split.err; echo "split rc=$?"
cat $O/f32.jsonl $O/split.jsonl | cut -c1-330; tail -3 $O/*.err
