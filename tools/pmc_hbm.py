#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into the per-kernel
HBM traffic table bench.py reads (profiles/*_hbm_traffic.csv).
usage: tools/pmc_hbm.py <fetch_dir> <write_dir> <out.csv> "<command that was profiled>"
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reads half of a wide coalesced stream
(MI355X_MICROARCH.md, HBM / rocprofv3 section); counters are in KiB."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def load(d, counter):
    acc, cnt = defaultdict(float), defaultdict(int)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {d}")
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            acc[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return acc, cnt


def main():
    fdir, wdir, out, cmd = sys.argv[1:5]
    fs, fc = load(fdir, "FETCH_SIZE")
    ws, wc = load(wdir, "WRITE_SIZE")
    rows = []
    for k in sorted(fs, key=lambda k: -(2 * fs[k] + ws.get(k, 0.0))):
        n = fc[k]
        rows.append((k, n, fs[k], ws.get(k, 0.0), (2 * fs[k] + ws.get(k, 0.0)) * 1024 / n / 2 ** 20))
    conv = [r for r in rows if r[0].startswith("conv_gemm_kernel")]  # (bench.py adds the pw_stream_kernel and ff320_kernel rows itself)
    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- {cmd}\n")
        fh.write("# hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reads half of a wide coalesced stream "
                 "(MI355X_MICROARCH.md, HBM)\n")
        fh.write("kernel,launches,fetch_size_kb_sum,write_size_kb_sum,hbm_mb_per_launch_corrected\n")
        if conv:
            n = sum(r[1] for r in conv)
            f, w = sum(r[2] for r in conv), sum(r[3] for r in conv)
            fh.write(f'"conv_gemm_kernel (all instantiations)",{n},{f:.0f},{w:.0f},{(2 * f + w) * 1024 / n / 2 ** 20:.2f}\n')
        for k, n, f, w, mb in rows:
            fh.write(f'"{k}",{n},{f:.0f},{w:.0f},{mb:.2f}\n')
    print(open(out).read()[:1500])


if __name__ == "__main__":
    main()
