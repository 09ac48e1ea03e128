#!/usr/bin/env python3
"""Fit the conv dispatch cost model to tools/sweep_small.sh logs and report its regret.

model: t(cfg, S) = ceil(tiles * S / slots(cfg)) * (a[cfg] + b[cfg] * KT / S [+ e[cfg] if S > 1]) + (S > 1) * (r0 + r1 * M * N * S)
The fitted constants go into pick_plan() of ccnet_amd/csrc/gemm_conv.hip.
usage: fit_dispatch.py log [log ...]
"""
import math
import re
import sys

import numpy as np
from scipy.optimize import least_squares

TILE = {0: (128, 128, 2), 19: (128, 128, 1), 15: (256, 128, 1), 3: (128, 320, 1), 23: (256, 320, 1), 20: (256, 256, 1)}
SPLIT = {3: 3, 15: 15, 19: 19, 23: 23}
CFGS = sorted(TILE)


def parse(path):
    rows = {}
    S = None
    for l in open(path):
        if l.startswith("== split-K"):
            S = int(l.split()[-1])
            continue
        m = re.match(r"(.{32}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) (.*)  best", l)
        if not m:
            continue
        name = m.group(1).strip()
        M, N, K = int(m.group(2)), int(m.group(3)), int(m.group(4))
        flops = 2.0 * M * N * K
        key = (name, M, N, K)
        d = rows.setdefault(key, {})
        for c, tf in re.findall(r"c(-?\d+)p?:\s*(\d+)", m.group(5)):
            c, tf = int(c), float(tf)
            if tf <= 0:
                continue
            t = flops / tf / 1e12 * 1e6
            if c == -1:
                if S == 0:
                    d[("h", 0)] = t
                continue
            if S == 0 or c not in TILE:
                continue
            if "GEGLU" in name and c not in (0, 20):
                continue
            if S > 1 and c not in SPLIT:
                continue
            s = S
            while s > 1 and (K // 64) // s < 13:
                s -= 1
            if s != S:
                continue  # the library lowered the factor: a duplicate of another row
            d[(c, S)] = t
    return rows


def tiles(M, N, c):
    bm, bn, _ = TILE[c]
    return math.ceil(M / bm) * math.ceil(N / bn)


def model(p, M, N, K, c, S, geglu):
    i = CFGS.index(c)
    a, b, e = p[3 * i:3 * i + 3]
    r0, r1, g = p[-3:]
    slots = 256 * TILE[c][2]
    rounds = math.ceil(tiles(M, N, c) * S / slots)
    t = rounds * (a * (g if geglu else 1.0) + b * (K / 64) / S + (e if S > 1 else 0.0))
    if S > 1:
        t += r0 + r1 * M * N * S * 1e-6
    return t


def main():
    data = []
    per_log = []
    for path in sys.argv[1:]:
        rows = parse(path)
        per_log.append((path, rows))
        for (name, M, N, K), d in rows.items():
            for (c, S), t in d.items():
                if c != "h":
                    data.append((M, N, K, c, S, "GEGLU" in name, t))
    x0 = np.array([8.0, 0.15, 4.0] * len(CFGS) + [5.0, 1.0, 1.2])

    def resid(p):
        return [math.log(max(model(p, M, N, K, c, S, g), 1e-3) / t) for M, N, K, c, S, g, t in data]

    sol = least_squares(resid, x0, bounds=(0, np.inf))
    p = sol.x
    r = np.array(resid(p))
    print(f"{len(data)} points, rms log error {np.sqrt((r ** 2).mean()):.3f}, max {np.abs(r).max():.3f}")
    for i, c in enumerate(CFGS):
        print(f"cfg {c:2d}: a {p[3 * i]:.2f} us  b {p[3 * i + 1]:.4f} us/kstep  e {p[3 * i + 2]:.2f} us")
    print(f"reduce r0 {p[-3]:.2f} us, r1 {p[-2]:.3f} us/M fp32, geglu factor {p[-1]:.2f}")
    for path, rows in per_log:
        th = tb = tm = 0.0
        for (name, M, N, K), d in rows.items():
            cand = {k: v for k, v in d.items() if k[0] != "h"}
            if not cand:
                continue
            best = min(cand.values())
            pick = min(cand, key=lambda k: model(p, M, N, K, k[0], k[1], "GEGLU" in name))
            h = d.get(("h", 0), float("nan"))
            th += h; tb += best; tm += cand[pick]
            flag = " <<" if cand[pick] > 1.1 * best else ""
            print(f"  {name:34s} M={M:6d} heur {h:6.1f} best {best:6.1f} model c{pick[0]}/S{pick[1]} {cand[pick]:6.1f}{flag}")
        print(f"{path}: heuristic {th:.0f} us, best {tb:.0f} us, model pick {tm:.0f} us")


if __name__ == "__main__":
    main()
