#!/bin/bash
# kernel-trace stats of one batch with the LayerNorm fold off and on (sequential streams), same box
root=$PWD; out=$root/gpurun_out/prof_fold
CMD="python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof"
export CCNET_OVERLAP=0
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  CCNET_LN_FOLD=$v timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$v -- $CMD > $out/stats$v.log 2>&1 || exit 1
  cp $(find $out/stats$v -name "*kernel_stats.csv" | head -1) $root/gpurun_out/fold${v}_kernel_stats.csv
  rm -rf $out/stats$v
done
