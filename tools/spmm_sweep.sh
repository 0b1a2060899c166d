for cfg in "64 512 2048" "32 512 2048" "128 512 2048" "64 512 256" "64 512 1" "64 2048 2048" "16 512 2048"; do set -- $cfg
echo "== SHORT_MAX=$1 LONG_MIN=$2 CHUNK=$3"; EGNN_SHORT_ROW_MAX=$1 EGNN_LONG_ROW_MIN=$2 EGNN_PLAN_CHUNK=$3 python tools/kernel_bench.py --only spmm --quick 2>&1 | grep -E '"gcn_sum_val"|sage_mean' | cut -c1-200; done
