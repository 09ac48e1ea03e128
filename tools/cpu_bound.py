#!/usr/bin/env python3
"""How much of a batch's wall time is CPU enqueue time (launch-bound check)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ccnet_amd.sampling import Sampler
from ccnet_amd.schedulers import DDIMScheduler
from ccnet_amd.synthetic import make_batch
dev = torch.device("cuda:0")
cn, un, vae = bench.build_models(dev, True)
sch = DDIMScheduler(clip_sample=False, **bench.SCHED); sch.set_timesteps(50)
s = Sampler(cn, un, sch, vae, guidance_scale=7.0, conditioning_scale=1.0, scale_factor=0.1)
b = {k: v.to(dev) for k, v in make_batch(list(range(16)), seed=42, acq=True).items()}
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = s.sample(b["noise"], b["cond"], b["prompt_embeds"], b["empty_prompt_embeds"], b.get("acquisition_time"))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {t1 - t0:.3f} s, total {t2 - t0:.3f} s", flush=True)
