#!/bin/bash
# Re-check of plan_conv's rules against forced (tile configuration, split-K) pairs on the layers where an alternative is plausible
# (tools/bench_conv.py --cold: TFLOP/s per configuration; c-1 = the rule's own pick).  usage: tools/sweep_plan.sh > out.txt
F='^(L2|U1|L3|U0|mid|L1 (proj|o-proj|ff2)|up )'
for S in 1 2 3 4; do
  echo "== split-K $S"
  timeout -k 10 280 python3 tools/bench_conv.py --cold --iters 12 --cfgs=-1,23,29,15,19,3,10 --splitk $S --filter "$F" 2>&1 | grep -E "best|failed" || exit 1
done
