#!/usr/bin/env python3
"""Throughput of the ControlNet training step (ccnet_amd.training.train_step: forward of both networks, backward through the
UNet decoder and the whole ControlNet, AdamW) at the SD-2.1-base geometry on one MI355X, synthetic latents.
Usage: python tools/train_throughput.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import G1, G1_CE, SCHED, build_models  # noqa: E402
from ccnet_amd import schedulers, training  # noqa: E402
from ccnet_amd.synthetic import make_batch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cn, un, _ = build_models(dev, True)
    sch = schedulers.DDPMScheduler(**SCHED)
    opt = torch.optim.AdamW(cn.parameters(), lr=1e-5)
    for B in [int(v) for v in sys.argv[1:]] or [1, 4, 8]:
        b = make_batch(list(range(B)), seed=42, acq=True)
        g = torch.Generator().manual_seed(1)
        e = torch.randn(B, 4, 64, 64, generator=g).to(dev)
        t = torch.randint(0, 1000, (B,), generator=g).to(dev)
        args = (e, b["noise"].to(dev), t, b["cond"].to(dev), b["prompt_embeds"].to(dev), b["acquisition_time"].to(dev).reshape(1, B))
        for _ in range(2):
            loss, _ = training.train_step(cn, un, sch, opt, *args, clip_grad_norm_by=1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            loss, _ = training.train_step(cn, un, sch, opt, *args, clip_grad_norm_by=1.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"batch {B:3d}: {dt * 1e3:8.1f} ms / step  {B / dt:6.2f} images/s  loss {loss.item():.4f}  "
              f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
