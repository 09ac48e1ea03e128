#!/usr/bin/env python3
"""Which Python lines of one eager sampler step issue torch copies (copy_, cat, clone, contiguous, to): the device-to-device
copies that ride in the captured graph beside the library's kernels.  usage: python3 tools/trace_copies.py [batch]"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda:0")
    from ccnet_amd.sampling import Sampler
    from ccnet_amd.schedulers import DDIMScheduler
    from ccnet_amd.synthetic import make_batch
    cn, un, vae = bench.build_models(dev, True)
    sch = DDIMScheduler(clip_sample=False, **bench.SCHED)
    sch.set_timesteps(50)
    s = Sampler(cn, un, sch, vae, guidance_scale=7.0, conditioning_scale=1.0, scale_factor=0.1, use_graph=False)
    b = {k: v.to(dev) for k, v in make_batch(list(range(B)), seed=42, acq=True).items()}
    st = s.prepare(b["cond"], b["prompt_embeds"], b["empty_prompt_embeds"], b.get("acquisition_time"))
    x = b["noise"].float().contiguous()
    x, _ = s.step(st, x, int(sch.timesteps[0]), None, index=0)  # warm (packs, caches)
    counts = collections.Counter()

    def wrap(name, fn):
        def inner(*a, **k):
            fr = [f for f in traceback.extract_stack()[:-1] if "ccnet_amd" in f.filename]
            t = a[0] if name != "cat" else a[0][0]
            r = fn(*a, **k)
            changed = name in ("cat", "clone", "copy_") or (isinstance(r, torch.Tensor) and r.data_ptr() != t.data_ptr())
            if fr and changed and isinstance(t, torch.Tensor) and t.is_cuda:
                counts[(name, os.path.basename(fr[-1].filename), fr[-1].lineno, tuple(t.shape))] += 1
            return r
        return inner
    torch.cat = wrap("cat", torch.cat)
    for n in ("copy_", "clone", "contiguous", "to", "float"):
        setattr(torch.Tensor, n, wrap(n, getattr(torch.Tensor, n)))
    s.step(st, x, int(sch.timesteps[1]), None, index=1)
    torch.cuda.synchronize()
    for (n, f, ln, shp), c in sorted(counts.items(), key=lambda kv: (kv[0][1], kv[0][2])):
        print(f"{c:4d} x {n:10s} {f}:{ln}  {shp}")
    print("total", sum(counts.values()))


if __name__ == "__main__":
    main()
