#!/usr/bin/env python3
"""Micro-benchmark of the fused feed-forward kernel (ccnet_ff_geglu_fused_bf16) against the three launches it replaces
(LayerNorm, ff.linear1 + GEGLU, ff.linear2 + residual) at the 64x64 level of the G1 geometry (CFG batch 32: 131072 rows x 320).
Random data, events on the launch stream, inputs rotated over several copies so that rows come from HBM as in the real step."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--copies", type=int, default=6)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    C_, Hd, M = 320, 1280, args.rows
    xs = [torch.randn(M, C_, generator=g).to(torch.bfloat16).to(dev) for _ in range(args.copies)]
    gamma, beta = (1 + 0.2 * torch.randn(C_, generator=g)).to(dev), (0.3 * torch.randn(C_, generator=g)).to(dev)
    w1 = (torch.randn(2 * Hd, C_, generator=g) / math.sqrt(C_)).to(dev)
    b1 = (0.5 * torch.randn(2 * Hd, generator=g)).to(dev)
    w2 = (torch.randn(C_, Hd, generator=g) * (0.5 / math.sqrt(Hd))).to(dev)
    b2 = (0.5 * torch.randn(C_, generator=g)).to(dev)
    pack = ops.ff_pack(w1, b1, w2)
    w1i, b1i = ops.geglu_interleave(w1.to(torch.bfloat16)).contiguous(), ops.geglu_interleave(b1).contiguous()
    w2b = w2.to(torch.bfloat16).contiguous()
    flops = 2.0 * M * C_ * (2 * Hd) + 2.0 * M * Hd * C_

    def fused(x):
        return ops.ff_fused(x, C_, pack, b2, gamma, beta, 1e-5)

    def chain(x):
        t = ops.layernorm(x, C_, gamma, beta)
        gl = ops.linear(t, w1i, b1i, 2 * Hd, C_, act=2)
        return ops.linear(gl, w2b, b2, C_, Hd, res1=x)

    for name, fn in (("fused", fused), ("three launches", chain), ("fused", fused), ("three launches", chain)):
        for i in range(3):
            fn(xs[i % args.copies])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters):
            fn(xs[i % args.copies])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        print(f"{name:15s} rows {M}: {us:8.1f} us  {flops / us * 1e-6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
