#!/usr/bin/env python3
"""Micro-benchmark of the fp32 edge convolutions (conv_in_small 4 -> 320 at 64x64, CFG batch 32; conv_out_small 320 -> 4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops  # noqa: E402
from ccnet_amd.ops import Act  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(16, 4, 64, 64, device=dev, generator=g)
    w = torch.randn(320, 4, 3, 3, device=dev, generator=g) / 6
    b = torch.randn(320, device=dev, generator=g)
    res = Act(torch.randn(16 * 4096, 320, device=dev, generator=g).to(torch.bfloat16), 16, 64, 64, 320)
    print(f"conv_in_small 4->320 @64 N=32:        {timeit(lambda: ops.conv_in_small(x, 32, w, b, 320, 320)):8.1f} us")
    print(f"conv_in_small 4->320 @64 N=32 +res:   {timeit(lambda: ops.conv_in_small(x, 32, w, b, 320, 320, res=res)):8.1f} us")
    c = torch.randn(16, 1, 512, 512, device=dev, generator=g)
    w1 = torch.randn(16, 1, 3, 3, device=dev, generator=g) / 3
    b1 = torch.randn(16, device=dev, generator=g)
    print(f"conv_in_small 1->16(64) @512 N=16:    {timeit(lambda: ops.conv_in_small(c, 16, w1, b1, 16, 64, act=1)):8.1f} us")
    xa = Act(torch.randn(32 * 4096, 320, device=dev, generator=g).to(torch.bfloat16), 32, 64, 64, 320)
    wo = (torch.randn(4, 9, 320, device=dev, generator=g) / 54).to(torch.bfloat16)
    bo = torch.randn(4, device=dev, generator=g)
    print(f"conv_out_small 320->4 @64 N=32:       {timeit(lambda: ops.conv_out_small(xa, wo, bo, 4)):8.1f} us")


if __name__ == "__main__":
    main()
