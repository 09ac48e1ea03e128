#!/usr/bin/env python3
"""Generates tests/golden/tiny_golden.npz from the fp32 oracle (oracle/, CPU).

The reference holds no fixtures for this path and its third-party model code cannot be imported here
(SURVEY.md section 8(c)), so these vectors pin the build's OWN restatement against drift ("parity unpinned"
with respect to upstream).  Inputs are stored explicitly; weights are regenerated from seeds by
ccnet_amd.synthetic.init_weights (a checksum of every state dict is stored and verified by the tests).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _common import SCHED, TINY, TINY_CE, TINY_VAE  # noqa: E402


def checksum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def build_oracles():
    import oracle.nets as on
    from ccnet_amd.synthetic import init_weights
    cn = on.ControlNet(**TINY, **TINY_CE)
    un = on.DiffusionModelUNet(out_channels=4, **TINY)
    vae = on.AutoencoderKL(**TINY_VAE)
    on.adjust_class_embedding(cn)
    on.adjust_class_embedding(un)
    init_weights(cn, 11)
    init_weights(un, 12)
    init_weights(vae, 13)
    return cn.eval(), un.eval(), vae.eval()


def main():
    import oracle.schedulers as osch
    from oracle.sampler import sample as osample, to_uint8
    torch.manual_seed(0)
    cn, un, vae = build_oracles()
    g = torch.Generator().manual_seed(1234)
    B = 2
    x = torch.randn(B, 4, 16, 16, generator=g)
    ctx = 0.5 * torch.randn(B, 77, 64, generator=g)
    empty = 0.5 * torch.randn(1, 77, 64, generator=g)
    cond = torch.rand(B, 1, 128, 128, generator=g) * 2 - 1
    acq = torch.tensor([100.0, 650.0])
    t = torch.tensor([500])
    with torch.no_grad():
        down, mid = cn(x, t, cond, 0.9, ctx, acq.reshape(B, 1))
        out = un(x, t, ctx, acq.reshape(B, 1), down, mid)
        plain = un(x, t, ctx, acq.reshape(B, 1))
    sch = osch.DDIMScheduler(clip_sample=False, **SCHED)
    sch.set_timesteps(8)
    res = osample(cn, un, sch, x, cond, ctx, empty, guidance_scale=3.0, conditioning_scale=1.0, acquisition_time=acq,
                  stage1=vae, scale_factor=0.5, record_latents=True)
    dsch = osch.DDPMScheduler(**SCHED)
    dsch.set_timesteps(8)
    dn = [torch.randn(B, 4, 16, 16, generator=g) for _ in range(8)]
    dres = osample(cn, un, dsch, x, cond, ctx, empty, guidance_scale=3.0, conditioning_scale=1.0, acquisition_time=acq,
                   ddpm_noise=dn, record_latents=True)
    arrays = {
        "x": x, "ctx": ctx, "empty": empty, "cond": cond, "acq": acq,
        "cn_down0": down[0], "cn_down_last": down[-1], "cn_mid": mid, "unet_out": out, "unet_plain": plain,
        "ddim_latents": torch.stack(res["latents"]), "ddim_image": res["image"],
        "ddim_u8": torch.from_numpy(to_uint8(res["image"])),
        "ddpm_noise": torch.stack(dn), "ddpm_latents": torch.stack(dres["latents"]),
        "ck_cn": torch.tensor(checksum(cn.state_dict())), "ck_un": torch.tensor(checksum(un.state_dict())),
        "ck_vae": torch.tensor(checksum(vae.state_dict())),
        "acp": sch.alphas_cumprod[[0, 1, 20, 500, 960, 980, 995, 999]],
        "ddim_ts": sch.timesteps,
    }
    path = os.path.join(ROOT, "tests", "golden", "tiny_golden.npz")
    np.savez_compressed(path, **{k: v.detach().cpu().numpy() for k, v in arrays.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
