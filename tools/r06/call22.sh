#!/bin/bash
set +e
for cfg in "default 1" "explicit 1" "default 0"; do set -- $cfg
  echo "== INST=$1 EVAL=$2"; INST=$1 EVAL=$2 timeout 600 python tools/r06/memset_real.py 2>&1 | grep -E "^#|^step|Error|error" | cut -c1-200 | head -12
done
