#!/bin/bash
# same-box A/B of an environment switch on the whole pipeline: tools/ab_envvar.sh CCNET_ATTN_PRESCALE 0 1 "16 1"
V=$1; A=$2; B=$3
python3 bench.py --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1  # clocks up
for N in ${4:-"16"}; do
  for X in $A $B $A $B; do
    v=$(env $V=$X timeout -k 10 200 python3 bench.py --batch $N --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>&1 | grep -o '"value": [0-9.]*') || exit 1
    echo "batch $N $V=$X $v"
  done
done
