#!/usr/bin/env python3
"""Phase breakdown of the conv kernel from the diagnostic (-DCCNET_STAMP) build: per-block s_memtime stamps at
entry / set-up done / first tile landed / main loop done / epilogue done.  Usage:
CCNET_HIP_LIB=build_ab/lib_stamp.so python tools/stamp_one.py N H C1 Cout R cfg [res] [act]"""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import lib, ops  # noqa: E402
from ccnet_amd.ops import Act  # noqa: E402

N, H, C1, Cout, R, cfg = [int(v) for v in sys.argv[1:7]]
with_res = len(sys.argv) > 7 and int(sys.argv[7])
act = int(sys.argv[8]) if len(sys.argv) > 8 else 0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
x = Act(torch.randn(N * H * H, C1, device=dev, generator=g).to(torch.bfloat16), N, H, H, C1)
K = R * R * C1
ws = [(torch.randn(Cout, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16) for _ in range(8)]
b = torch.randn(Cout, device=dev, generator=g)
res = Act(torch.randn(N * H * H, Cout // (2 if act == 2 else 1), device=dev, generator=g).to(torch.bfloat16), N, H, H, Cout) \
    if with_res else None
l = lib.load()
l.ccnet_set_option(0, cfg)
dbg = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
l.ccnet_debug_stamp_buffer.argtypes = [C.c_void_p]
import time
t_end = time.perf_counter() + float(os.environ.get("STAMP_WARM_S", "2.0"))  # let the clocks settle under this load
it = 0
while time.perf_counter() < t_end:
    for _ in range(16):
        ops.conv(x, ws[it % 8], b, Cout, R=R, res1=res, act=act)
        it += 1
    torch.cuda.synchronize()
l.ccnet_debug_stamp_buffer(dbg.data_ptr())
dbg.zero_()
ops.conv(x, ws[5], b, Cout, R=R, res1=res, act=act)
torch.cuda.synchronize()
l.ccnet_debug_stamp_buffer(None)
d = dbg.cpu().numpy().reshape(-1, 8)
d = d[d[:, 0] != 0]
t = d[:, :5].astype(np.float64)
ph = np.diff(t, axis=1)
names = ["setup", "first-tile wait", "main loop", "epilogue"]
print(f"blocks {len(d)}  kernel span {(t[:, 4].max() - t[:, 0].min()):.0f} ticks")
for i, nm in enumerate(names):
    print(f"  {nm:16s} mean {ph[:, i].mean():9.0f}  p10 {np.percentile(ph[:, i], 10):9.0f}  p90 {np.percentile(ph[:, i], 90):9.0f}")
print(f"  per-block total  mean {(t[:, 4] - t[:, 0]).mean():9.0f}")
# in-kernel clock over the main loop: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime), slots 5 / 6
rt = (d[:, 6] - d[:, 5]).astype(np.float64)
ok = rt > 0
if ok.any():
    clk = (t[ok, 3] - t[ok, 2]) / rt[ok] * 0.1
    print(f"  main-loop clock  median {np.median(clk):.3f} GHz  p10 {np.percentile(clk, 10):.3f}  p90 {np.percentile(clk, 90):.3f}")

# persistent launches: stamps 2..4 exist per tile (slot = virtual block); look at lockstep between the CUs
full = dbg.cpu().numpy().reshape(-1, 8)
rows = np.nonzero(full[:, 4] != 0)[0]
if len(rows) > len(d):
    G = 256
    t0 = full[rows, 2].min()
    print(f"tiles stamped {len(rows)}; per round of {G} tiles: start-of-loop / end-of-loop / end-of-epilogue, ticks from t0")
    for r in range(0, min(len(rows) // G, 8)):
        sel = rows[(rows >= r * G) & (rows < (r + 1) * G)]
        a2, a3, a4 = (full[sel, k].astype(np.float64) - t0 for k in (2, 3, 4))
        print(f"  round {r}: loop start {a2.mean():8.0f} +- {a2.std():6.0f}   loop {np.mean(a3 - a2):7.0f}   "
              f"epilogue {np.mean(a4 - a3):7.0f} (p10 {np.percentile(a4 - a3, 10):6.0f} p90 {np.percentile(a4 - a3, 90):6.0f})   "
              f"epilogue-end spread (std) {a4.std():6.0f}")
