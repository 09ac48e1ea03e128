#!/bin/bash
# kernel-trace stats of the training step (G1, batch $1): tools/prof_train.sh 8
root=$PWD; out=$root/gpurun_out/prof_train
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/tools/train_throughput.py ${1:-8} > $out/stats.log 2>&1 || exit 1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $root/gpurun_out/train_kernel_stats.csv
rm -rf $out/stats
