#!/bin/bash
# small-batch regimes: forced tile configuration x split-K factor over the UNet shapes; usage: sweep_small.sh <cfg batch> [S list]
IM=${1:-2}; SL=${2:-"0 1 2 4 8"}
for S in $SL; do
  echo "== split-K $S"
  timeout -k 10 250 python3 tools/bench_conv.py --images $IM --cold --iters 20 --cfgs=-1,0,19,15,3,23,20 --splitk $S --filter "@" 2>&1 | grep -v -E "amdgpu|differs|TOTAL|VAE" || exit 1
done
