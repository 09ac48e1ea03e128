#!/bin/bash
# same-box A/B of a library option over batch sizes: tools/ab_opt.sh "6=3" "1 2 4"   (CCNET_HIP_OPTIONS, see ccnet_set_option)
OPT=$1
python3 bench.py --batch 2 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1  # clocks up
for B in ${2:-"1 4 16"}; do
  for O in "" "$OPT" "" "$OPT"; do
    v=$(CCNET_HIP_OPTIONS="$O" timeout -k 10 200 python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>&1 | grep -o '"value": [0-9.]*') || exit 1
    echo "batch $B options '$O' $v"
  done
done
