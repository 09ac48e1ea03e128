"""Timing of LayerNorm + linear (two launches) against the one-launch form with the statistics taken in the streaming kernel.
   python tools/bench_ln_inline.py   (rows = 131072: batch 16 x CFG at 64x64)"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccnet_amd import ops

dev = torch.device("cuda:0")
M = 131072
g = torch.Generator().manual_seed(1)
x = torch.randn(M, 320, generator=g).to(torch.bfloat16).to(dev)
gamma, beta = (1 + 0.1 * torch.randn(320, generator=g)), 0.1 * torch.randn(320, generator=g)


def timeit(f, reps=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


gm0, bt0 = gamma.to(dev), beta.to(dev)
print(f"layernorm alone {timeit(lambda: ops.layernorm(x, 320, gm0, bt0)):.1f} us, row_stats alone {timeit(lambda: ops.row_stats(x, 320)):.1f} us", flush=True)
for Cout in (320, 960):
    w = torch.randn(Cout, 320, generator=g) / math.sqrt(320)
    wp = w.to(torch.bfloat16).to(dev)
    wf, bf, cs = (t.to(dev) for t in ops.fold_layernorm(w, None, gamma, beta))
    gm, bt = gamma.to(dev), beta.to(dev)
    two = timeit(lambda: ops.linear(ops.layernorm(x, 320, gm, bt), wp, None, Cout, 320))
    lin = timeit(lambda: ops.linear(x, wp, None, Cout, 320))
    one = timeit(lambda: ops.linear(x, wf, bf, Cout, 320, colsum=cs))
    print(f"Cout {Cout}: layernorm + linear {two:.1f} us (linear alone {lin:.1f} us), statistics in the kernel {one:.1f} us", flush=True)
