#!/usr/bin/env python3
"""GroupNorm (+SiLU) kernels on the G1 shapes at CFG batch 32: time and algorithmic GB/s (one read + one write, bf16) per shape,
inputs rotated through > 512 MB so they come from HBM as in the real step.  usage: tools/bench_gn.py [--iters 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops  # noqa: E402
from ccnet_amd.ops import Act  # noqa: E402

# (name, N, H, C1, C2, launches per denoising step over UNet + ControlNet, roughly)
SHAPES = [
    ("64x64 320", 32, 64, 320, 0, 14), ("64x64 640+320", 32, 64, 640, 320, 1), ("64x64 320+320", 32, 64, 320, 320, 2),
    ("64x64 640", 32, 64, 640, 0, 1), ("32x32 320", 32, 32, 320, 0, 2), ("32x32 640", 32, 32, 640, 0, 13),
    ("32x32 1280+640", 32, 32, 1280, 640, 1), ("32x32 640+640", 32, 32, 640, 640, 1), ("32x32 640+320", 32, 32, 640, 320, 1),
    ("32x32 1280", 32, 32, 1280, 0, 1), ("16x16 640", 32, 16, 640, 0, 2), ("16x16 1280", 32, 16, 1280, 0, 13),
    ("16x16 1280+1280", 32, 16, 1280, 1280, 2), ("16x16 1280+640", 32, 16, 1280, 640, 1), ("8x8 1280", 32, 8, 1280, 0, 12),
    ("8x8 1280+1280", 32, 8, 1280, 1280, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    tot_ms = 0.0
    for name, N, H, C1, C2, per_step in SHAPES:
        C = C1 + C2
        nbytes = N * H * H * C * 2
        ncopy = max(2, min(16, -(-(512 << 20) // nbytes)))
        xs = [Act(torch.randn(N * H * H, C1, device=dev, generator=g).to(torch.bfloat16), N, H, H, C1) for _ in range(ncopy)]
        x2s = [Act(torch.randn(N * H * H, C2, device=dev, generator=g).to(torch.bfloat16), N, H, H, C2) for _ in range(ncopy)] if C2 else None
        gamma = torch.randn(C, device=dev, generator=g)
        beta = torch.randn(C, device=dev, generator=g)
        t_end = time.perf_counter() + 0.05
        while time.perf_counter() < t_end:
            ops.groupnorm(xs[0], gamma, beta, 32, 1e-5, True, x2s[0] if C2 else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(args.iters):
            ops.groupnorm(xs[it % ncopy], gamma, beta, 32, 1e-5, True, x2s[it % ncopy] if C2 else None)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.iters * 1e3
        tot_ms += us * per_step * 50 / 1e3
        print(f"{name:18s} {2 * nbytes / 2 ** 20:7.1f} MB r+w  {us:7.1f} us  {2 * nbytes / us / 1e3:6.0f} GB/s   x{per_step:2d} per step", flush=True)
    print(f"sum over a 50-step batch (launch counts approximate): {tot_ms:.0f} ms")


if __name__ == "__main__":
    main()
