#!/usr/bin/env python3
"""Known-byte-count launches for calibrating FETCH_SIZE on the conv kernel's two LDS-DMA patterns (run under
`rocprofv3 --pmc FETCH_SIZE`): M = 131072 rows x K = 320 against ONE column tile of weights (Cout = 256), so every input
line is needed exactly once (84 MB of x + 0.16 MB of weights per launch) whatever the caches do.
  * cfg 10: 256x256 tile, K-step 64 -> DMA rows of 128 B (one full line per 8 lanes)
  * cfg 20: 256x256 tile, K-step 32 -> DMA rows of 64 B (half a line per 4 lanes; the other half one K-step later)
Each configuration is launched ITERS times on fresh inputs (rotated over > 256 MiB so nothing survives in the Infinity Cache)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import lib, ops  # noqa: E402
from ccnet_amd.ops import Act  # noqa: E402

ITERS = 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
M, K, Cout = 131072, 320, 256
xs = [Act(torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16), 32, 64, 64, K) for _ in range(ITERS)]
w = (torch.randn(Cout, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
b = torch.randn(Cout, device=dev, generator=g)
l = lib.load()
for cfg in (10, 20):
    l.ccnet_set_option(0, cfg)
    for x in xs:
        ops.conv(x, w, b, Cout, R=1, pad=0)
    torch.cuda.synchronize()
print(f"per launch: x {M * K * 2 / 2 ** 20:.1f} MiB + w {Cout * K * 2 / 2 ** 20:.2f} MiB read, out {M * Cout * 2 / 2 ** 20:.1f} MiB written")
