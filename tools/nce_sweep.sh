#!/bin/bash
echo "== fwd waves=2 (prebuilt)"; python tools/kernel_bench.py --only nce 2>&1 | grep '"nce"' | cut -c1-230
python -m pytest tests -m gpu -q -k "nce" -p no:cacheprovider 2>&1 | tail -2
echo "== rebuild fwd waves=1"; EGNN_EXTRA_FLAGS="-DEGNN_NCE_FWD_WAVES=1" python efficient-gnns_amd/build.py --force > /tmp/b.log 2>&1; tail -1 /tmp/b.log
python tools/kernel_bench.py --only nce 2>&1 | grep '"nce"' | cut -c1-230
