#!/bin/bash
# kernel-trace stats of one batch with the LayerNorm fold on (sequential streams), beside the default
root=$PWD; out=$root/gpurun_out/prof_fold
CMD="python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof"
export CCNET_OVERLAP=0
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
CCNET_LN_FOLD=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/stats.log 2>&1 || exit 1
cd $root
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) gpurun_out/fold1_kernel_stats.csv
rm -rf $out/stats
