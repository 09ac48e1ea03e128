#!/bin/bash
# same-box A/B of an environment switch over batch sizes: tools/ab_env.sh CCNET_SKIP_AUX "1 4 16"
VAR=$1
python3 bench.py --batch 2 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1  # clocks up
for B in ${2:-"1 4 16"}; do
  for O in 0 1 0 1; do
    v=$(env $VAR=$O timeout -k 10 200 python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>&1 | grep -o '"value": [0-9.]*') || exit 1
    echo "batch $B $VAR=$O $v"
  done
done
