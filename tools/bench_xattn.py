#!/usr/bin/env python3
"""Micro-benchmark of the fused cross-attention kernel (ccnet_xattn_fused_bf16) against the three launches it replaces
(LayerNorm-folded to_q, 77-key flash kernel, to_out + residual) at the 64x64 level of the G1 geometry (CFG batch 32: 131072 rows x
320, 5 heads of 64, 77 keys).  Random data, events on the launch stream, inputs rotated over several copies (rows from HBM)."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--copies", type=int, default=6)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    C_, HW, S, B = 320, 4096, 77, args.images
    M = B * HW
    xs = [torch.randn(M, C_, generator=g).to(torch.bfloat16).to(dev) for _ in range(args.copies)]
    gamma, beta = (1 + 0.2 * torch.randn(C_, generator=g)).to(dev), (0.3 * torch.randn(C_, generator=g)).to(dev)
    wq = (torch.randn(C_, C_, generator=g) * (2.0 / math.sqrt(C_))).to(dev)
    wo = (torch.randn(C_, C_, generator=g) / math.sqrt(C_)).to(dev)
    bo = (0.5 * torch.randn(C_, generator=g)).to(dev)
    kv = torch.randn(B * S, 2 * C_, generator=g).to(torch.bfloat16).to(dev)
    pack = ops.xattn_pack(wq, None, wo, 5)
    img = ops.xattn_kv_image(kv, B, S, C_)
    qs = math.log2(math.e) / 8.0
    wqi = ops.fold_layernorm(wq.float() * qs, None, gamma, beta)
    wqs = (wq * qs).to(torch.bfloat16).contiguous()
    wob = wo.to(torch.bfloat16).contiguous()
    flops = 2.0 * M * C_ * (2 * C_ + 2 * S)

    def fused(x):
        return ops.xattn_fused(x, C_, 5, pack, bo, gamma, beta, img, S, HW, 1e-5)

    def chain(x):
        w, b, cs = wqi
        q = ops.linear(x, w, b, C_, C_, colsum=cs)
        o = ops.attention_d64(q, kv[:, :C_], kv[:, C_:], B, 5, HW, S, 1.0 / 8.0, prescaled=True)
        return ops.linear(o, wob, bo, C_, C_, res1=x)

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
