#!/bin/bash
# SQ counters of the streaming pointwise kernel on the ff.linear1 + GEGLU / qkv shapes of the 64x64 level: tools/pmc_pw.sh <tag>
tag=$1
R=$PWD
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  out=$R/gpurun_out/${tag}_pw_s$i
  mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $out -- python3 $R/tools/bench_conv.py --filter "L0 (qkv|ff1\+GEGLU)" --iters 3 > $out.log 2>&1) || { tail -5 $out.log; exit 1; }
  python3 tools/pmc_sq.py $out gpurun_out/${tag}_pw_s$i.csv "bench_conv L0 qkv / ff1+GEGLU, counter set $i" > /dev/null
  grep -E "^#|kernel|pw_stream" gpurun_out/${tag}_pw_s$i.csv
done
