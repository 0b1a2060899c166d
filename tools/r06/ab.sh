#!/bin/bash
# A/B of two builds of libegnn_hip.so on ONE box: tools/r06/libegnn_hip_{a,b}.so, interleaved A B A B
set +e
for v in a b a b; do
  cp tools/r06/libegnn_hip_$v.so efficient-gnns_amd/lib/libegnn_hip.so
  echo "== build $v"; bash tools/r06/quick_bench.sh | head -6
done
