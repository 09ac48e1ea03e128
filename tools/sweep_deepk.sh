#!/bin/bash
# deep-K small-M layers (one image): split-K factors beyond 8; needs a build with kMaxSplit raised
for S in 8 12 16 24 32; do
  echo "== split-K $S"
  timeout -k 10 250 python3 tools/bench_conv.py --images 2 --cold --iters 20 --cfgs=0,19,15 --splitk $S --filter "conv3x3 1280|conv3x3 2560|ff2 5120|U2 conv|U0 skip|L1 conv3x3|L1 ff2|U3 conv" 2>&1 | grep -v -E "amdgpu|differs|TOTAL|VAE" || exit 1
done
