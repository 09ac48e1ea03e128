#!/bin/bash
# L2 -> fabric read traffic (FETCH_SIZE, gfx950-corrected x2) of the 16x16-level ff.linear1 launches (8192 x 10240 x 1280, plain and
# with the GEGLU epilogue) under the N-fastest tile walk (--order 0) and the windowed walk (--order 2).  usage: tools/pmc_raster_fetch.sh <tag>
tag=$1; R=$PWD
for ord in 0 2; do
  out=$R/gpurun_out/${tag}_raster_fetch_o$ord; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $R/tools/bench_conv.py --filter "L2 ff1" --iters 4 --order $ord > $out.log 2>&1) || exit 1
  python3 - $out $ord <<'PY'
import csv, glob, sys, collections
d, ord_ = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_gemm_kernel<4, 2, 2, 5" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[("GEGLU" if "5, true" in r["Kernel_Name"] else "plain", r["Grid_Size"])].append(float(r["Counter_Value"]))
for g, v in sorted(acc.items()):
    print(f"order {ord_} {g[0]:5s} grid_size {g[1]:>8s}: {len(v):3d} launches, FETCH_SIZE corrected {2 * sum(v) / len(v) / 1024:8.1f} MB per launch (algorithmic reads 47.2 MB)")
PY
done
