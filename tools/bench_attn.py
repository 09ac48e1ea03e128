#!/usr/bin/env python3
"""Micro-benchmark of the d=64 flash attention kernel on the G1 self / cross attention shapes at CFG batch 32."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops  # noqa: E402

SHAPES = [("L0 self 4096x4096 h5", 32, 5, 4096, 4096), ("L1 self 1024x1024 h10", 32, 10, 1024, 1024),
          ("L2 self 256x256 h20", 32, 20, 256, 256), ("L0 cross 4096x77 h5", 32, 5, 4096, 77)]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for name, B, H, Sq, Skv in SHAPES:
        if only and only not in name:
            continue
        C = H * 64
        q = torch.randn(B * Sq, C, device=dev, generator=g).to(torch.bfloat16)
        k = torch.randn(B * Skv, C, device=dev, generator=g).to(torch.bfloat16)
        v = torch.randn(B * Skv, C, device=dev, generator=g).to(torch.bfloat16)
        pre = os.environ.get("ATTN_PRE", "0") == "1"
        if pre:
            q = (q.float() * (0.125 * math.log2(math.e))).to(torch.bfloat16)
        for _ in range(2):
            ops.attention_d64(q, k, v, B, H, Sq, Skv, 0.125, prescaled=pre)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.attention_d64(q, k, v, B, H, Sq, Skv, 0.125, prescaled=pre)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        fl = 4.0 * B * H * Sq * Skv * 64
        print(f"{name:24s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
