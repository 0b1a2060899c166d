#!/bin/bash
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
echo "== gsp tests"; timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "gsp or gpw or pairwise" > gpurun_out/r06/t15.log 2>&1; echo rc=$?; tail -3 gpurun_out/r06/t15.log | cut -c1-300
for P in "" f32; do
echo "== bench gpw EGNN_GEMM_PIPE=$P"; EGNN_GEMM_PIPE=$P timeout 600 python bench.py --training gpw --steps 100 --warmup 3 --cpu-epochs 0 --no-local-roofline --reference-epochs 0 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['parity']['ok'], j['parity']['max_rel_err'], j['roofline_gsp'])"
done
