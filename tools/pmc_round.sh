#!/bin/bash
# PMC passes for the SpMM kernel (counters only with --kernel-trace, one counter per pass: TCC slot limits).
set +e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/pmc; mkdir -p /tmp/pmc $R/gpurun_out/pmc
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc/f -o fetch -- python $R/tools/spmm_pmc.py > $R/gpurun_out/pmc/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc/w -o write -- python $R/tools/spmm_pmc.py > $R/gpurun_out/pmc/write.log 2>&1; echo "write rc=$?"
cd $R
find /tmp/pmc -name "*counter_collection.csv" | head; for f in $(find /tmp/pmc -name "*counter_collection.csv"); do head -3 $f | cut -c1-400; cp $f gpurun_out/pmc/$(basename $f); done
python tools/parse_pmc.py /tmp/pmc gpurun_out/pmc/spmm_traffic.json
