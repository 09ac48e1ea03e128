"""Diagnostic: first op whose per-image output depends on the image's position in the batch (checksums of raw bits)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import SCHED, build_models  # noqa: E402
from ccnet_amd import ops  # noqa: E402
from ccnet_amd.sampling import Sampler  # noqa: E402
from ccnet_amd.schedulers import DDIMScheduler  # noqa: E402
from ccnet_amd.synthetic import make_batch  # noqa: E402

dev = torch.device("cuda:0")
cn, un, vae = build_models(dev, True)
B = 16
Bc = 2 * B
idx = list(range(B))
perm = [(7 * i + 3) % B for i in range(B)]
log = []


def bits_sum(t, parts):
    t = t.contiguous()
    v = t.view(torch.int16 if t.element_size() == 2 else torch.int32).to(torch.int64)
    return v.reshape(parts, -1).sum(1).cpu()


def rec(name, out, extra=""):
    outs = out if isinstance(out, tuple) else (out,)
    for k, o in enumerate(outs):
        t = o.t if isinstance(o, ops.Act) else o
        if not torch.is_tensor(t):
            continue
        rows = t.shape[0]
        if isinstance(o, ops.Act):
            t = t[:, :o.C]
        if isinstance(o, ops.Act) and o.N != Bc:
            log.append((f"{name}[{k}] {tuple(t.shape)} N={o.N} (skipped)", None))
        elif rows % Bc == 0 and (isinstance(o, ops.Act) or rows == Bc):
            log.append((f"{name}[{k}] {tuple(t.shape)} {extra}", bits_sum(t, Bc)))
        elif rows % Bc == 0 and name == "linear_f32":  # [T * Bc, n]: row t * Bc + b
            T = rows // Bc
            log.append((f"{name}[{k}] table {tuple(t.shape)}", bits_sum(t.reshape(T, Bc, -1).transpose(0, 1), Bc)))
        elif rows % B == 0 and rows // B in (1,) :
            log.append((f"{name}[{k}] B-rows {tuple(t.shape)}", torch.cat([bits_sum(t, B)] * 2)))
        else:
            log.append((f"{name}[{k}] {tuple(t.shape)} (unchunked)", None))


def wrap(name):
    f = getattr(ops, name)

    def g(*a, **k):
        out = f(*a, **k)
        extra = ""
        if name == "conv":
            extra = f"Cout={a[3]} R={k.get('R', 3)} res1={'res1' in k and k['res1'] is not None} x2={k.get('x2') is not None}"
        rec(name, out, extra)
        return out
    setattr(ops, name, g)


for nm in ("conv", "groupnorm", "layernorm", "attention_d64", "conv_in_small", "conv_out_small", "linear_f32"):
    wrap(nm)


def run(ids):
    log.clear()
    sch = DDIMScheduler(clip_sample=False, **SCHED)
    sch.set_timesteps(50)
    s = Sampler(cn, un, sch, None, guidance_scale=7.0, conditioning_scale=1.0, scale_factor=0.1, overlap=False, use_graph=False)
    b = {k: v.to(dev) for k, v in make_batch(ids, seed=42, acq=True).items()}
    s.sample(b["noise"], b["cond"], b["prompt_embeds"], b["empty_prompt_embeds"], b["acquisition_time"], max_steps=1, decode=False)
    return list(log)


A = run(idx)
P = run(perm)
assert len(A) == len(P)
pos = torch.tensor([perm.index(i) for i in idx])
pos = torch.cat([pos, pos + B])
shown = 0
for n, ((na, ca), (np_, cp)) in enumerate(zip(A, P)):
    if ca is None or cp is None:
        continue
    if ca.numel() != Bc:
        continue
    bad = (cp[pos] != ca).nonzero().flatten().tolist()
    if bad:
        print(f"op {n}: {na}: differing CFG-batch rows (original order) {bad}")
        shown += 1
        if shown >= 12:
            break
print("ops recorded:", len(A), "first differing shown:", shown)
