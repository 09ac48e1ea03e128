#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/pmc_calibrate.py per conv instance (KiB per launch) -> gpurun_out/r02_fetch_calibration.txt
root=$PWD; out=$root/gpurun_out/pmc_cal
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 $root/tools/pmc_calibrate.py > $out/$c.log 2>&1 || exit 1
done
cd $root
python3 - <<'PY' > gpurun_out/r02_fetch_calibration.txt
import csv, glob, collections, re
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_cal/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "conv_gemm_kernel" in r["Kernel_Name"]:
                acc[re.search(r"conv_gemm_kernel<[^>]*>", r["Kernel_Name"]).group(0)].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{c} {k}: launches {len(v)}, KiB per launch mean {sum(v)/len(v):.0f} min {min(v):.0f} max {max(v):.0f} "
              f"= {sum(v)/len(v)/1024:.1f} MiB (algorithmic: 80.2 MiB read, 64.0 MiB written)")
PY
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
