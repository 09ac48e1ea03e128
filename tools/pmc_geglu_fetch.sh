#!/bin/bash
# L2 -> fabric read traffic of the L0 ff.linear1 + GEGLU launch (131072 x 2560 x 320) by tile walk: persistent (256 blocks) vs one
# block per tile, N-fastest vs M-fastest order.  usage: tools/pmc_geglu_fetch.sh <tag>
tag=$1; R=$PWD
for ord in 0 1; do
  out=$R/gpurun_out/${tag}_geglu_fetch_o$ord; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $R/tools/bench_conv.py --filter "L0 ff1\+GEGLU" --cfgs 23 --persist 1,0 --iters 4 --order $ord > $out.log 2>&1) || exit 1
  python3 - $out $ord <<'PY'
import csv, glob, sys, collections
d, ord_ = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_gemm_kernel<4, 2, 2, 5, true" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Grid_Size"]].append(float(r["Counter_Value"]))
for g, v in sorted(acc.items()):
    print(f"order {ord_} grid_size {g:>8s}: {len(v):3d} launches, FETCH_SIZE {sum(v) / len(v) / 1024:8.1f} MB raw = {2 * sum(v) / len(v) / 1024:8.1f} MB corrected (algorithmic input 85.5 MB)")
PY
done
