#!/bin/bash
# Row ranges of the many-group streaming launches (L1 ff.linear1 + GEGLU: 20 cout groups; L1 qkv: 8): time and L2 -> fabric
# reads per launch by ccnet_set_option key 11.  usage: tools/pw_ranges.sh
R=$PWD
for rg in ${RANGES:-64 48 32 24}; do
  export CCNET_HIP_OPTIONS=11=$rg
  echo "== ranges $rg"
  timeout -k 10 120 python3 tools/bench_conv.py --cold --iters 20 --cfgs=-1,-1 --filter "L1 ff1\+GEGLU|L1 qkv" 2>&1 | grep -E "^L1" | cut -c1-110 || exit 1
  out=$R/gpurun_out/pw_ranges_$rg; mkdir -p $out
  (cd /tmp && export TMPDIR=/tmp && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $R/tools/bench_conv.py --filter "L1 ff1\+GEGLU|L1 qkv" --cfgs=-1 --iters 4 > $out.log 2>&1) || exit 1
  python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pw_stream_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[(r["Kernel_Name"][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
for g, v in sorted(acc.items()):
    print(f"   {g[0]} grid {g[1]:>8s}: {len(v):3d} launches, fetch {2 * sum(v) / len(v) / 1024:8.1f} MB corrected")
PY
  rm -rf $out
done
