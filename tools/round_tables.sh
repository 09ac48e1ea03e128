#!/bin/bash
# numbers of the final build on one box: images/s over batch sizes, AEKL batch 64 (configs[3]), training step: tools/round_tables.sh <tag>
tag=$1
python3 bench.py --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1  # clocks up
for B in 1 2 4 8 16 32; do
  v=$(timeout -k 10 300 python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>/dev/null | grep -o '"value": [0-9.]*') || exit 1
  echo "batch $B $v" | tee -a gpurun_out/${tag}_batch_table.log
done
timeout -k 10 300 python3 tools/aekl_throughput.py 2>/dev/null | tee -a gpurun_out/${tag}_batch_table.log || exit 1
timeout -k 10 400 python3 tools/train_throughput.py 2>/dev/null | tee gpurun_out/${tag}_train_throughput.log
