#!/bin/bash
# A/B of the LayerNorm fold with the two-stream overlap off (sequential kernels), same box
for v in 0 1 0 1; do echo "== CCNET_LN_FOLD=$v CCNET_OVERLAP=0"; CCNET_OVERLAP=0 CCNET_LN_FOLD=$v timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
