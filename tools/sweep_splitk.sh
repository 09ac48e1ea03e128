#!/bin/bash
# sweep forced tile configuration x split-K factor over the small-M deep-K layers (8x8 and 16x16 levels)
for S in 1 2 3 4 6 8; do
  echo "== split-K $S"
  timeout -k 10 200 python3 tools/bench_conv.py --cold --iters 16 --cfgs=-1,23,19,3,27,15 --splitk $S --filter "$1" 2>&1 | grep -v -E "amdgpu|differs|TOTAL" || exit 1
done
