#!/usr/bin/env python3
"""BASELINE.json configs[3]: AEKL stage1 encode / decode throughput, batch 64 at 512^2 (seeded random weights, chunked batches)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
_, _, vae = bench.build_models(dev, True)
g = torch.Generator().manual_seed(0)
x = (torch.rand(64, 1, 512, 512, generator=g) * 2 - 1).to(dev)
z = torch.randn(64, vae.latent_channels if hasattr(vae, "latent_channels") else 4, 64, 64, generator=g).to(dev)
for name, fn, arg in (("decode", vae.decode, z), ("encode", vae.encode, x)):
    fn(arg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn(arg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    o = out[0] if isinstance(out, tuple) else out
    assert torch.isfinite(o).all()
    print(f"AEKL {name}: batch 64 @512^2 in {dt * 1e3:.1f} ms = {64 / dt:.0f} images/s")
