#!/bin/bash
# One round's committed evidence: kernel-trace stats + the two PMC passes of the same command (run on the GPU box).
# usage: tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/{stats,fetch,write}, summaries under gpurun_out/
tag=$1; root=$PWD; out=$root/gpurun_out/prof_$tag
CMD="python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof"
# per-kernel durations and counters are taken with the ControlNet / UNet-encoder overlap off (isolated kernels)
export CCNET_OVERLAP=0
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/stats.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $CMD --no-graph > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $CMD --no-graph > $out/write.log 2>&1 || exit 1
cd $root
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
python3 tools/pmc_hbm.py $out/fetch $out/write gpurun_out/${tag}_hbm_traffic.csv "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof --no-graph (counter collection crashes on graph replay; same kernels, same order)"
# the raw traces are large: keep only the summaries
rm -rf $out/stats $out/fetch $out/write
