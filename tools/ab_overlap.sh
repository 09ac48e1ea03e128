#!/bin/bash
# same-box A/B of the ControlNet / UNet-encoder overlap: tools/ab_overlap.sh "1 2 4 8 16"
python3 bench.py --batch 2 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1  # clocks up
for B in ${1:-"1 2 4 8 16"}; do
  for O in 0 1 0 1; do
    v=$(CCNET_OVERLAP=$O timeout -k 10 200 python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>&1 | grep -o '"value": [0-9.]*') || exit 1
    echo "batch $B overlap $O $v"
  done
done
