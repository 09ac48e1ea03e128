#!/bin/bash
# Idle time between consecutive kernels of the captured step (sequential form: one stream): rocprofv3 --kernel-trace of one batch,
# then the sum of (start[i+1] - end[i]) over the launches.  usage: tools/gap_analysis.sh <tag>
tag=$1; root=$PWD; out=$root/gpurun_out/gaps_$tag
export CCNET_OVERLAP=0
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof > $out.log 2>&1 || exit 1
cd $root
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
# the last 50 steps' worth: take the final 60 % of launches (the graph replays), skip set-up
n = len(rows); rows = rows[int(n * 0.4):]
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 <= g < 20000]
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} launches over {span / 1e6:.1f} ms: kernels {busy / 1e6:.1f} ms, gaps < 20 us: {sum(small) / 1e6:.1f} ms ({len(small)} gaps, mean {sum(small) / max(len(small), 1) / 1e3:.2f} us), "
      f"larger gaps {sum(g for g in gaps if g >= 20000) / 1e6:.1f} ms, overlaps {sum(-g for g in gaps if g < 0) / 1e6:.1f} ms")
big = sorted(((g, i) for i, g in enumerate(gaps) if g >= 20000), reverse=True)
print(f"{len(big)} gaps >= 20 us; the 12 largest (us, kernel before -> kernel after):")
for g, i in big[:12]:
    print(f"  {g / 1e3:9.1f}  {rows[i][2][:60]} -> {rows[i + 1][2][:60]}")
import collections
c = collections.Counter((rows[i][2][:50], rows[i + 1][2][:50]) for g, i in big)
for (a, b), n in c.most_common(6):
    tot = sum(g for g, i in big if (rows[i][2][:50], rows[i + 1][2][:50]) == (a, b))
    print(f"  {n:4d} x {tot / n / 1e3:8.1f} us  {a} -> {b}")
PY
rm -rf $out
