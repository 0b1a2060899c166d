#!/bin/bash
echo "== BK=16 (prebuilt)"; python tools/kernel_bench.py --only gemm,nce 2>&1 | grep -E '"gemm"|"nce"' | cut -c1-230
echo "== rebuild BK=32"; EGNN_EXTRA_FLAGS="-DEGNN_BK=32" python efficient-gnns_amd/build.py --force > /tmp/build32.log 2>&1; tail -2 /tmp/build32.log
python tools/kernel_bench.py --only gemm,nce 2>&1 | grep -E '"gemm"|"nce"' | cut -c1-230
python -m pytest tests -m gpu -q -k "gemm or nce or gsp or linear" -p no:cacheprovider 2>&1 | tail -2
