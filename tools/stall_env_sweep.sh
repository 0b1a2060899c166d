#!/bin/bash
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null); nproc $(nproc)"
python -c "import torch; print('torch threads', torch.get_num_threads(), torch.get_num_interop_threads())"
run() { echo "=== $*"; env "$@" timeout 300 python tools/time_sharded.py 2>&1 | grep -E "total mean|^backward|^eval|^fwd model|^nce fwd" | sed 's/  */ /g' | tr '\n' ';'; echo; }
run OMP_NUM_THREADS=1 MKL_NUM_THREADS=1
run OMP_NUM_THREADS=4
run A=1
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
