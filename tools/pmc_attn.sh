#!/bin/bash
# SQ counters of the attention instances on the 4096x4096 self-attention shape: tools/pmc_attn.sh <tag>
tag=$1
R=$PWD
for m in 1 2 3; do
  out=$R/gpurun_out/${tag}_attn_m$m
  mkdir -p $out
  export CCNET_HIP_OPTIONS=7=$m
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out -- python3 $R/tools/bench_attn.py 3 "L0 self" > $out.log 2>&1) || exit 1
  python3 tools/pmc_sq.py $out gpurun_out/${tag}_attn_m$m.csv "bench_attn mode $m" > /dev/null
  grep attn_d64 gpurun_out/${tag}_attn_m$m.csv
done
