"""Diagnostic: does a sample's result depend on its position in the batch (G1, batch 16)?  Per-step latent differences
between a batch and the same indices in another order, and between two identical runs, with the stream overlap on / off."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bench import G1, G1_CE, SCHED, STAGE1, build_models  # noqa: E402
from ccnet_amd.sampling import Sampler  # noqa: E402
from ccnet_amd.schedulers import DDIMScheduler  # noqa: E402
from ccnet_amd.synthetic import make_batch  # noqa: E402

dev = torch.device("cuda:0")
cn, un, vae = build_models(dev, True)
steps = int(os.environ.get("STEPS", "4"))
B = int(os.environ.get("B", "16"))
idx = list(range(B))
perm = [(7 * i + 3) % B for i in range(B)]
inv = torch.tensor([perm.index(i) for i in idx], device=dev)


def run(s, ids):
    b = {k: v.to(dev) for k, v in make_batch(ids, seed=42, acq=True).items()}
    return s.sample(b["noise"], b["cond"], b["prompt_embeds"], b["empty_prompt_embeds"], b["acquisition_time"],
                    record_latents=True, max_steps=steps, decode=False)["latents"]


for kw in (dict(overlap=True), dict(overlap=False), dict(overlap=False, use_graph=False)):
    sch = DDIMScheduler(clip_sample=False, **SCHED)
    sch.set_timesteps(50)
    s = Sampler(cn, un, sch, None, guidance_scale=7.0, conditioning_scale=1.0, scale_factor=0.1, **kw)
    a, b_, p = run(s, idx), run(s, idx), run(s, perm)
    print(kw, "same-batch rerun per-step max diff:", " ".join(f"{(x - y).abs().max().item():.2e}" for x, y in zip(a, b_)))
    print(kw, "permuted per-step max diff:       ", " ".join(f"{(x[inv] - y).abs().max().item():.2e}" for x, y in zip(p, a)))
    d = (p[0][inv] - a[0]).abs().flatten(1).max(1).values
    print("   step-0 per-index diff:", " ".join(f"{v:.1e}" for v in d.tolist()))
