import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ccnet_amd import ops
cuda = torch.device("cuda:0")
for rows, C, with_res in [(70000, 320, True), (8192, 1280, False), (8192, 1280, True), (32768, 640, True)]:
    g = torch.Generator().manual_seed(5 + rows + C)
    x = (torch.randn(rows, C, generator=g)).to(torch.bfloat16).to(cuda)
    w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(torch.bfloat16).to(cuda)
    b = (0.5 * torch.randn(C, generator=g) + 2.0).to(cuda)
    res = ((torch.randn(rows, C, generator=g) * (0.2 + 3.0 * torch.rand(rows, 1, generator=g))
            + 3.0 * torch.randn(rows, 1, generator=g)).to(torch.bfloat16).to(cuda)) if with_res else None
    out, st = ops.linear(x, w, b, C, C, res1=res, emit_stats=True)
    of = out.float()
    P = st.shape[1]; span = C // P
    for p in range(P):
        seg = of[:, p * span:(p + 1) * span]
        em = (st[:, p, 0] - seg.mean(1)).abs()
        ev = (st[:, p, 1] / span - seg.var(1, unbiased=False)).abs() / seg.var(1, unbiased=False)
        print(rows, C, with_res, "part", p, "mean err max", em.max().item(), "var rel err max", ev.max().item(), "argmax row", ev.argmax().item(),
              "frac bad", (ev > 5e-3).float().mean().item())
