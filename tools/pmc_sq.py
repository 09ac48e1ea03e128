#!/usr/bin/env python3
"""Aggregate one rocprofv3 --pmc pass of SQ counters per kernel (sums over dispatches) and derive the ratios the kernel work
needs: MFMA busy share, VALU / LDS issue share, LDS bank-conflict share.
usage: tools/pmc_sq.py <pmc_dir> <out.csv> "<command that was profiled>"
Ratios (all of summed counters): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES): SQ_BUSY_CYCLES counts per shader
engine (8 CUs x 4 SIMDs = 32 matrix pipes on MI355X: 256 CUs, 8 XCDs x 4 SEs), the MFMA counter per SIMD, so this is the share of
matrix-pipe cycles that were busy while the kernel ran (issue cycles, an upper bound of the useful-flop share);
lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; valu_active, lds_active, wait_lds relative to SQ_WAVE_CYCLES."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    d, out, cmd = sys.argv[1:4]
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {d}")
    first = None
    for f in files:
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            c = row["Counter_Name"]
            first = first or c
            acc[name][c] += float(row["Counter_Value"])
            if c == first:
                cnt[name] += 1
    counters = sorted({c for v in acc.values() for c in v})

    def ratio(v, a, b):
        return v.get(a, 0.0) / v[b] if v.get(b) else float("nan")

    with open(out, "w") as fh:
        fh.write(f"# rocprofv3 --pmc {' '.join(counters)} -- {cmd}\n")
        fh.write("kernel,launches," + ",".join(counters) + ",mfma_pipe_busy_share,valu_active_per_wave_cycle,"
                 "lds_active_per_wave_cycle,wait_lds_per_wave_cycle,lds_bank_conflict_per_lds_active\n")
        for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0.0)):
            v = acc[k]
            fh.write(f'"{k}",{cnt[k]},' + ",".join(f"{v.get(c, 0.0):.0f}" for c in counters) +
                     f',{ratio(v, "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES") / 32:.4f}'
                     f',{ratio(v, "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"):.4f}'
                     f',{ratio(v, "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"):.4f}'
                     f',{ratio(v, "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"):.4f}'
                     f',{ratio(v, "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):.4f}\n')
    print("".join(open(out).readlines()[:14]))


if __name__ == "__main__":
    main()
