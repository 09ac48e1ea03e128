#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernels (GroupNorm, LayerNorm, fp32 embedding linear) on G1 shapes at CFG
batch 32: algorithmic GB/s (1 read + 1 write of bf16 per element; weight bytes for the fp32 linear)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops  # noqa: E402
from ccnet_amd.ops import Act  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    N = 32
    for C, H in [(320, 64), (640, 64), (960, 64), (640, 32), (1920, 32), (1280, 16), (2560, 16), (1280, 8), (2560, 8)]:
        x = Act(torch.randn(N * H * H, C, device=dev, generator=g).to(torch.bfloat16), N, H, H, C)
        gm, bt = torch.randn(C, device=dev), torch.randn(C, device=dev)
        ms = timeit(lambda: ops.groupnorm(x, gm, bt, 32, 1e-5, True))
        gb = x.t.numel() * 4 / 1e9
        print(f"groupnorm+silu C={C:5d} H={H:3d}: {ms * 1e3:8.1f} us  {gb / ms * 1e3:7.1f} GB/s (algorithmic 4 B/elt)", flush=True)
    for rows, C in [(131072, 320), (32768, 640), (8192, 1280), (2048, 1280)]:
        x = torch.randn(rows, C, device=dev, generator=g).to(torch.bfloat16)
        gm, bt = torch.randn(C, device=dev), torch.randn(C, device=dev)
        ms = timeit(lambda: ops.layernorm(x, C, gm, bt))
        gb = x.numel() * 4 / 1e9
        print(f"layernorm rows={rows:7d} C={C:5d}: {ms * 1e3:8.1f} us  {gb / ms * 1e3:7.1f} GB/s", flush=True)
    for B, K, Nout in [(32, 1280, 20160), (32, 1280, 9600), (32, 1280, 1280), (32, 320, 1280)]:
        x = torch.randn(B, K, device=dev, generator=g)
        w = torch.randn(Nout, K, device=dev, generator=g) / K ** 0.5
        b = torch.randn(Nout, device=dev, generator=g)
        ms = timeit(lambda: ops.linear_f32(x, w, b))
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        err = (ops.linear_f32(x, w, b).double() - ref).abs().max().item()
        print(f"linear_f32 B={B} K={K} N={Nout}: {ms * 1e3:8.1f} us  {w.numel() * 4 / 1e9 / ms * 1e3:7.1f} GB/s  "
              f"max err {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
