#!/usr/bin/env python3
"""Summarise a tools/sweep_small.sh log: per shape the heuristic's time (split-K 0 row, cfg -1) and the best forced pairs."""
import collections
import re
import sys

best = collections.defaultdict(dict)
S = None
for l in open(sys.argv[1]):
    if l.startswith("== split-K"):
        S = int(l.split()[-1])
        continue
    m = re.match(r"(.{32}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) (.*)  best", l)
    if not m:
        continue
    name = m.group(1).strip()
    flops = 2.0 * int(m.group(2)) * int(m.group(3)) * int(m.group(4))
    for c, tf in re.findall(r"c(-?\d+)p?:\s*(\d+)", m.group(5)):
        if float(tf) > 0:
            best[name][(int(c), S)] = flops / float(tf) / 1e12 * 1e6
tot_h = tot_b = 0.0
for name, d in best.items():
    h = d.get((-1, 0), float("nan"))
    items = sorted(((k, v) for k, v in d.items() if k != (-1, 0)), key=lambda kv: kv[1])[:3]
    tot_h += h
    tot_b += min(h, items[0][1])
    print(f"{name:34s} heuristic {h:7.1f} us   best: " + " ".join(f"c{c}/S{s}:{t:.1f}" for (c, s), t in items))
print(f"sum heuristic {tot_h:.0f} us, sum best {tot_b:.0f} us")
