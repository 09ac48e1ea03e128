#!/bin/bash
# L2 -> fabric read traffic (FETCH_SIZE, gfx950-corrected x2) and time of the 3x3 convolutions under the two K walks
# (library option 10: 0 = (tap, channel chunk), 1 = (channel chunk, tap)) and, for the split-K layers of the 16x16 / 8x8 levels,
# under the N-fastest / M-fastest tile walks (--order 0 / 1).  usage: tools/pmc_korder_fetch.sh <tag>
tag=$1; R=$PWD
summ() {  # <dir> <label>
  python3 - $1 "$2" <<'PY'
import csv, glob, sys, collections
d, label = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_gemm" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[(r["Kernel_Name"].split("conv_gemm")[1][:40], r["Grid_Size"])].append(float(r["Counter_Value"]))
for g, v in sorted(acc.items()):
    print(f"{label} conv_gemm{g[0]:40s} grid {g[1]:>8s}: {len(v):3d} launches, FETCH_SIZE corrected {2 * sum(v) / len(v) / 1024:8.1f} MB per launch")
PY
}
for ko in 0 1; do
  out=$R/gpurun_out/${tag}_korder_fetch_k$ko; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && CCNET_HIP_OPTIONS=10=$ko timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $R/tools/bench_conv.py --filter "^(L0|U3|L1|U2) conv3x3" --iters 4 > $out.log 2>&1) || exit 1
  summ $out "korder $ko"
  rm -rf $out
done
for ord in 0 1; do
  out=$R/gpurun_out/${tag}_korder_fetch_o$ord; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $R/tools/bench_conv.py --filter "^(L2|U1|L3|U0) conv3x3" --iters 4 --order $ord > $out.log 2>&1) || exit 1
  summ $out "order $ord"
  rm -rf $out
done
