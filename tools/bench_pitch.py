"""Does the row pitch of the OUTPUT matter to the streaming pointwise kernel (L2 channel spread of the 32 rows a wave stores)?
   python tools/bench_pitch.py   (rows = 131072)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccnet_amd import ops
from ccnet_amd.ops import Act

dev = torch.device("cuda:0")
M = 131072
g = torch.Generator().manual_seed(1)
x = Act(torch.randn(M, 320, generator=g).to(torch.bfloat16).to(dev), 32, 64, 64, 320)


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


for name, Cout, act in (("qkv 320->960", 960, 0), ("ff1+GEGLU 320->2560", 2560, 2), ("to_q 320->320", 320, 0)):
    w = (torch.randn(Cout, 320, generator=g) / math.sqrt(320)).to(torch.bfloat16).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    Co = Cout // 2 if act == 2 else Cout
    for pad in (0, 64, 128, 192, 256):
        buf = torch.empty(M, Co + pad, dtype=torch.bfloat16, device=dev)
        out = Act(buf[:, :Co], 32, 64, 64, Co)
        t = timeit(lambda: ops.conv(x, w, b, Cout, R=1, pad=0, act=act, out=out))
        print(f"{name}: output pitch {2 * (Co + pad)} B = {2 * (Co + pad) / 128:g} lines: {t:.1f} us", flush=True)
