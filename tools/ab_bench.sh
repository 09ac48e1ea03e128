#!/bin/bash
# whole-pipeline A/B of library tuning options on one box: tools/ab_bench.sh "" "5=0" "4=0" ...  (each arg = CCNET_HIP_OPTIONS)
for o in "$@"; do
  echo "== options '$o'"
  CCNET_HIP_OPTIONS="$o" timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"kernel_ms_total": [0-9.]*' | tr '\n' ' ' || exit 1
  echo
done
