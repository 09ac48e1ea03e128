#!/bin/bash
# GroupNorm paths on the G1 shapes: option 6 = 4 (multi-group one-pass kernel off) vs default
for o in "6=4" ""; do echo "== CCNET_HIP_OPTIONS='$o'"; CCNET_HIP_OPTIONS="$o" timeout -k 10 120 python tools/bench_mem.py 2>&1 | grep groupnorm; done
