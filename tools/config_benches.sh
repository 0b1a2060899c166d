#!/bin/bash
# Secondary BASELINE.json configs (parity-test cases; timed here for DESIGN.md, not the headline line)
mkdir -p gpurun_out
for cfg in "sage lpw" "gcn gpw" "gcn kd" "sage nce" "gcn supervised"; do set -- $cfg
echo "== $1 $2"; timeout 600 python bench.py --gnn $1 --training $2 --steps 10 --warmup 3 --cpu-epochs 0 2>&1 | tail -1 | python -c "
import json,sys
l=sys.stdin.readline()
try:
    d=json.loads(l); print(json.dumps({k:d[k] for k in ('value','ms_per_step','phases_ms','last_losses')}), d['config']['workload'][:100])
except Exception as e: print('FAILED', l[:300])
"; done
echo "== S=8192 (script default max_samples)"; timeout 600 python bench.py --steps 20 --warmup 3 --cpu-epochs 0 --max-samples 8192 2>&1 | tail -1 | cut -c1-400
