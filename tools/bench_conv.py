#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv/linear kernel on the G1 (SD-2.1) shapes at CFG batch 32.
Random data (never zeros: MI355X clocks higher on zero operands), timed with events on the launch stream."""
import argparse
import math
import os
import re
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ccnet_amd import ops  # noqa: E402
from ccnet_amd.ops import Act  # noqa: E402

SHAPES = [
    # name, N, H, C1, C2, Cout, R, stride, ups
    ("L0 conv3x3 320->320 @64", 32, 64, 320, 0, 320, 3, 1, 1),
    ("L0 proj 1x1 320->320 @64", 32, 64, 320, 0, 320, 1, 1, 1),
    ("L0 qkv 320->960 @64", 32, 64, 320, 0, 960, 1, 1, 1),
    ("L0 ff1 320->2560 @64", 32, 64, 320, 0, 2560, 1, 1, 1),
    ("L0 ff2 1280->320 @64", 32, 64, 1280, 0, 320, 1, 1, 1),
    ("U3 conv3x3 960->320 @64", 32, 64, 640, 320, 320, 3, 1, 1),
    ("L1 conv3x3 640->640 @32", 32, 32, 640, 0, 640, 3, 1, 1),
    ("L1 proj 1x1 640->640 @32", 32, 32, 640, 0, 640, 1, 1, 1),
    ("L1 ff1 640->5120 @32", 32, 32, 640, 0, 5120, 1, 1, 1),
    ("L1 ff2 2560->640 @32", 32, 32, 2560, 0, 640, 1, 1, 1),
    ("U2 conv3x3 1920->640 @32", 32, 32, 1280, 640, 640, 3, 1, 1),
    ("L2 conv3x3 1280->1280 @16", 32, 16, 1280, 0, 1280, 3, 1, 1),
    ("L2 ff1 1280->10240 @16", 32, 16, 1280, 0, 10240, 1, 1, 1),
    ("L2 ff2 5120->1280 @16", 32, 16, 5120, 0, 1280, 1, 1, 1),
    ("U1 conv3x3 2560->1280 @16", 32, 16, 1280, 1280, 1280, 3, 1, 1),
    ("L3 conv3x3 1280->1280 @8", 32, 8, 1280, 0, 1280, 3, 1, 1),
    ("U0 conv3x3 2560->1280 @8", 32, 8, 1280, 1280, 1280, 3, 1, 1),
    ("up conv3x3 1280->1280 8->16", 32, 8, 1280, 0, 1280, 3, 1, 2),
    ("down conv3x3 s2 320 64->32", 32, 64, 320, 0, 320, 3, 2, 1),
    ("L0 o-proj 320->320 +res @64", 32, 64, 320, 0, 320, 1, 1, 1, True),
    ("L1 o-proj 640->640 +res @32", 32, 32, 640, 0, 640, 1, 1, 1, True),
    ("L2 o-proj 1280->1280 +res @16", 32, 16, 1280, 0, 1280, 1, 1, 1, True),
    ("L2 qkv 1280->3840 @16", 32, 16, 1280, 0, 3840, 1, 1, 1),
    ("L2 proj 1x1 1280->1280 @16", 32, 16, 1280, 0, 1280, 1, 1, 1),
    ("L1 qkv 640->1920 @32", 32, 32, 640, 0, 1920, 1, 1, 1),
    ("mid ff2 5120->1280 +res @8", 32, 8, 5120, 0, 1280, 1, 1, 1, True),
    ("U0 skip 1x1 2560->1280 @8", 32, 8, 2560, 0, 1280, 1, 1, 1),
    ("L3 conv3x3 1280->1280 +res @8", 32, 8, 1280, 0, 1280, 3, 1, 1, True),
    ("L0 conv3x3 320->320 +res @64", 32, 64, 320, 0, 320, 3, 1, 1, True),
    ("L1 conv3x3 640->640 +res @32", 32, 32, 640, 0, 640, 3, 1, 1, True),
    ("L0 ff2 1280->320 +res @64", 32, 64, 1280, 0, 320, 1, 1, 1, True),
    ("L1 ff2 2560->640 +res @32", 32, 32, 2560, 0, 640, 1, 1, 1, True),
    ("L2 ff2 5120->1280 +res @16", 32, 16, 5120, 0, 1280, 1, 1, 1, True),
    ("mid 1x1 1280->1280 @8", 32, 8, 1280, 0, 1280, 1, 1, 1),
    ("L0 ff1+GEGLU 320->2560 @64", 32, 64, 320, 0, 2560, 1, 1, 1, False, 2),
    ("L1 ff1+GEGLU 640->5120 @32", 32, 32, 640, 0, 5120, 1, 1, 1, False, 2),
    ("L2 ff1+GEGLU 1280->10240 @16", 32, 16, 1280, 0, 10240, 1, 1, 1, False, 2),
    ("mid ff1+GEGLU 1280->10240 @8", 32, 8, 1280, 0, 10240, 1, 1, 1, False, 2),
    ("X960 conv3x3 960->960 @32", 32, 32, 960, 0, 960, 3, 1, 1),      # Cout divisible by 192 and 320 (tile experiments)
    ("X960 1x1 960->960 @64", 32, 64, 960, 0, 960, 1, 1, 1),
    ("X768 conv3x3 768->768 @32", 24, 32, 768, 0, 768, 3, 1, 1),      # M = 24576 = 96 * 256 = 128 * 192, Cout = 3 * 256 = 4 * 192
    ("VAE conv3x3 128->128 @512 N=4", 4, 512, 128, 0, 128, 3, 1, 1),
    ("VAE conv3x3 256->256 @256 N=8", 8, 256, 256, 0, 256, 3, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--prewarm-ms", type=float, default=30.0,
                    help="keep the GPU busy this long before a shape's first timed configuration (its clocks drop while the "
                         "shape's tensors are generated).  The first column of the first LARGE shape can still read ~10 %% low (fresh pool "
                         "memory): put a throw-away configuration first when that column matters")
    ap.add_argument("--filter", type=str, default="")
    ap.add_argument("--cfgs", type=str, default="-1", help="comma list of forced tile configs (-1 = heuristic)")
    ap.add_argument("--dbg-zero", type=int, default=0)
    ap.add_argument("--splitk", type=int, default=0, help="force the split-K factor (0 = heuristic)")
    ap.add_argument("--order", type=int, default=-1, help="force tile order: 0 N-fastest, 1 M-fastest")
    ap.add_argument("--persist", type=str, default="1", help="comma list: 1 persistent launches, 0 one block per tile")
    ap.add_argument("--scale", type=float, default=1.0, help="epilogue scale (diagnostic builds key experiments on it)")
    ap.add_argument("--images", type=int, default=0, help="override the CFG batch of every shape (2 = one image with CFG)")
    ap.add_argument("--cold", action="store_true", help="rotate over 12 weight copies so weights come from HBM")
    args = ap.parse_args()
    from ccnet_amd import lib as _lib
    cfgs = [(int(c), int(pp)) for c in args.cfgs.split(",") for pp in args.persist.split(",")]
    if args.dbg_zero:
        from ccnet_amd import lib
        lib.load().ccnet_set_option(1, args.dbg_zero)
    _lib.load().ccnet_set_option(2, args.order)
    _lib.load().ccnet_set_option(3, args.splitk)
    dev = torch.device("cuda:0")
    tot_f, tot_t = 0.0, 0.0
    rs_seen = False
    for shape in SHAPES:
        name, N, H, C1, C2, Cout, R, stride, ups = shape[:9]
        if args.images:
            N = args.images
        with_res = len(shape) > 9 and shape[9]
        act = shape[10] if len(shape) > 10 else 0
        if args.filter and not re.search(args.filter, name):
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        x1 = Act(torch.randn(N * H * H, C1, device=dev, generator=g).to(torch.bfloat16), N, H, H, C1)
        x2 = Act(torch.randn(N * H * H, C2, device=dev, generator=g).to(torch.bfloat16), N, H, H, C2) if C2 else None
        K = R * R * (C1 + C2)
        w = (torch.randn(Cout, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
        ncopy = min(24, max(1, -(-320 * 2 ** 20 // (w.numel() * 2)))) if args.cold else 1
        ws = [w] + [w.clone() for _ in range(ncopy - 1)]
        b = torch.randn(Cout, device=dev, generator=g)
        Ho = H * ups // stride
        res = Act(torch.randn(N * Ho * Ho, Cout, device=dev, generator=g).to(torch.bfloat16), N, Ho, Ho, Cout) \
            if with_res else None
        rs = []
        _lib.load().ccnet_set_option(0, -1)
        _lib.load().ccnet_set_option(4, 0)
        ref_out = ops.conv(x1, w, b, Cout, R=R, stride=stride, ups=ups, x2=x2, res1=res, act=act, scale=args.scale).t.clone()
        t_end = time.perf_counter() + args.prewarm_ms * 1e-3 * (1 if rs_seen else 10)  # (longer for the first shape of a process)
        rs_seen = True
        while time.perf_counter() < t_end:
            ops.conv(x1, w, b, Cout, R=R, stride=stride, ups=ups, x2=x2, res1=res, act=act, scale=args.scale)
        torch.cuda.synchronize()
        for cfg, pers in cfgs:
            _lib.load().ccnet_set_option(0, cfg)
            _lib.load().ccnet_set_option(4, pers)
            try:
                for _ in range(6):
                    o = ops.conv(x1, w, b, Cout, R=R, stride=stride, ups=ups, x2=x2, res1=res, act=act, scale=args.scale)
            except Exception as e:  # report and keep sweeping
                print(f"cfg {cfg} failed: {e}", flush=True)
                rs.append((0.0, 1e9))
                continue
            torch.cuda.synchronize()
            o_chk = ops.conv(x1, w, b, Cout, R=R, stride=stride, ups=ups, x2=x2, res1=res, act=act, scale=args.scale).t
            if not torch.equal(o_chk, ref_out):
                print(f"  !! cfg {cfg} differs from the heuristic config: max abs "
                      f"{(o_chk.float() - ref_out.float()).abs().max().item():.3e}", flush=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(args.iters):
                o = ops.conv(x1, ws[it % ncopy], b, Cout, R=R, stride=stride, ups=ups, x2=x2, res1=res, act=act, scale=args.scale)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            flops = 2.0 * o.rows * Cout * K
            rs.append((flops / ms / 1e9, ms))
        best = max(range(len(cfgs)), key=lambda i: rs[i][0])
        tot_f += flops
        tot_t += rs[best][1]
        print(f"{name:32s} M={o.rows:7d} N={Cout:5d} K={K:5d} " + " ".join(f"c{c[0]}{'p' if c[1] else ''}:{r[0]:6.0f}" for c, r in zip(cfgs, rs))
              + f"  best c{cfgs[best][0]}{'p' if cfgs[best][1] else ''} {rs[best][1] * 1e3:7.1f} us", flush=True)
    if tot_t:
        print(f"{'TOTAL of best (equal weight)':34s} {tot_f / tot_t / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
