#!/usr/bin/env python3
"""Run one conv shape with one forced tile config a few times (for rocprofv3 --pmc passes)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import lib, ops  # noqa: E402
from ccnet_amd.ops import Act  # noqa: E402

N, H, C1, Cout, R, cfg, iters = [int(v) for v in sys.argv[1:8]]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
x = Act(torch.randn(N * H * H, C1, device=dev, generator=g).to(torch.bfloat16), N, H, H, C1)
K = R * R * C1
ws = [(torch.randn(Cout, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16) for _ in range(8)]
b = torch.randn(Cout, device=dev, generator=g)
lib.load().ccnet_set_option(0, cfg)
for it in range(iters):
    ops.conv(x, ws[it % 8], b, Cout, R=R)
torch.cuda.synchronize()
