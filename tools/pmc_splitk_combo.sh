#!/bin/bash
# 16x16-level split-K 3x3 convolutions (8192 x 1280 x 11520 / 23040): time and L2 -> fabric read traffic (FETCH_SIZE, gfx950-corrected
# x2) under the K walk (library option 10) x the tile walk (--order 0 N-fastest / 1 M-fastest).  usage: tools/pmc_splitk_combo.sh <tag>
tag=$1; R=$PWD
for ko in 0 1; do for ord in 0 1; do
  echo "== korder $ko order $ord"
  CCNET_HIP_OPTIONS=10=$ko timeout -k 10 200 python3 tools/bench_conv.py --cold --iters 20 --order $ord --filter "^(L2|U1) conv3x3" 2>&1 | grep "us$" | awk '{printf "  %-34s %s us\n", substr($0,1,32), $(NF-1)}' || exit 1
  out=$R/gpurun_out/${tag}_combo_k${ko}_o$ord; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && CCNET_HIP_OPTIONS=10=$ko timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $R/tools/bench_conv.py --filter "^(L2|U1) conv3x3" --iters 4 --order $ord > $out.log 2>&1) || exit 1
  python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_gemm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("conv_gemm")[1][:44]].append(float(r["Counter_Value"]))
for g, v in sorted(acc.items()):
    print(f"  conv_gemm{g:44s}: {len(v):3d} launches, FETCH_SIZE corrected {2 * sum(v) / len(v) / 1024:8.1f} MB per launch")
PY
  rm -rf $out
done; done
