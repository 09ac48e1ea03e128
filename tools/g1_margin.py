#!/usr/bin/env python3
"""GPU side of the G1 parity-margin measurement: the HIP path at the SD-2.1-base geometry against the
fp32 oracle's outputs stored under tests/golden/g1_margin/ by tools/g1_margin_oracle.py (same seeded weights and inputs; the
oracle takes minutes per image on host cores, so it ran once, on the CPU, and its vectors are committed).

  ddim50      BASELINE.json configs[1]: indices 0..15 as ONE batch of 16, 50 DDIM steps, guidance 7.0, scale_factor 0.1:
              per-index latent max-abs, image PSNR (float, data range 2) and uint8 differences; per-step latent max-abs of
              indices 0 and 1.
  ddim200     the reference's canonical script shape (src/bash/INF_CTRL_CCNET_P1.sh:17,21,30): 200 DDIM steps, guidance 0.0 and
              None, scale_factor 0.01, indices 0 and 1.
  ddpm_tail   the last 50 DDPM steps (t = 49 .. 0), clip_sample=True, injected start latent and noise, indices 0 and 1.

Environment switches of the lowering (CCNET_LN_INLINE, CCNET_ATTN_PRESCALE, CCNET_FF_FUSED, CCNET_HIP_OPTIONS, ...) apply as
usual: running the tool once per switch gives the per-lowering error attribution (--tag names the run).
Writes one JSON (default gpurun_out/g1_margin_<tag>.json)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden", "g1_margin")

STAGE1 = dict(spatial_dims=2, in_channels=1, out_channels=1, num_channels=[128, 256, 256, 512], latent_channels=4,
              num_res_blocks=2, attention_levels=[False] * 4, with_encoder_nonlocal_attn=False,
              with_decoder_nonlocal_attn=False)


def product_nets(dev):
    """The product halves of tests/_common.build_pair (seeds 101 / 102 / 103, parameters drawn on the CPU generator)."""
    import ccnet_amd.nets as pn
    from _common import G1, G1_CE
    from ccnet_amd.synthetic import init_weights
    pc = pn.ControlNet(**{**G1, **G1_CE})
    pn.adjust_class_embedding(pc)
    init_weights(pc, 101)
    pu = pn.DiffusionModelUNet(out_channels=4, **G1)
    pn.adjust_class_embedding(pu)
    init_weights(pu, 102)
    pv = pn.AutoencoderKL(**STAGE1)
    init_weights(pv, 103)
    return pc.to(dev).eval(), pu.to(dev).eval(), pv.to(dev).eval()


def psnr(a, b, data_range=2.0):
    mse = ((a.double() - b.double()) ** 2).mean().item()
    return 10 * math.log10(data_range ** 2 / max(mse, 1e-20))


def compare(res, ref, k, i):
    """Image k of the run against entry i of a fixture."""
    from ccnet_amd.sampling import to_uint8
    lat = res["latent"][k].float().cpu()
    rl = torch.from_numpy(ref["latent"][i])
    img = res["image"][k:k + 1].float().cpu()
    ri = torch.from_numpy(ref["image"][i:i + 1].astype(np.float32))
    u8 = to_uint8(res["image"][k:k + 1]).cpu().int()[0]
    ru8 = torch.from_numpy(ref["u8"][i]).int()
    du8 = (u8 - ru8).abs()
    return {"latent_max_abs": (lat - rl).abs().max().item(), "latent_ref_max": rl.abs().max().item(),
            "psnr_db": psnr(img, ri), "psnr_u8_db": psnr(u8.float(), ru8.float(), 255.0),
            "u8_mean_abs": du8.float().mean().item(), "u8_max_abs": int(du8.max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=str, default="ddim50,ddim200,ddpm_tail")
    ap.add_argument("--tag", type=str, default="default")
    ap.add_argument("--out", type=str, default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    import ccnet_amd.schedulers as ps
    from _common import SCHED
    from ccnet_amd.sampling import Sampler
    from ccnet_amd.synthetic import make_batch
    from g1_margin_oracle import ddpm_tail_inputs
    t0 = time.time()
    pc, pu, pv = product_nets(dev)
    print(f"networks built in {time.time() - t0:.0f} s", flush=True)
    switches = {k: v for k, v in os.environ.items() if k.startswith("CCNET_")}
    report = {"tag": args.tag, "switches": switches}

    def run(s, b, **kw):
        return s.sample(b["noise"].to(dev), b["cond"].to(dev), b["prompt_embeds"].to(dev), b["empty_prompt_embeds"].to(dev),
                        b["acquisition_time"].to(dev), **kw)

    for case in args.cases.split(","):
        if case == "ddim50":
            refs = {}
            for g0 in range(0, 16, 4):
                path = os.path.join(GOLD, f"ddim50_gs7_idx{g0:02d}.npz")
                if os.path.exists(path):
                    z = np.load(path)
                    for i, idx in enumerate(z["indices"]):
                        refs[int(idx)] = (z, i)
            if not refs:
                continue
            sch = ps.DDIMScheduler(clip_sample=False, **SCHED)
            sch.set_timesteps(50)
            s = Sampler(pc, pu, sch, pv, guidance_scale=7.0, conditioning_scale=1.0, scale_factor=0.1)
            res = run(s, make_batch(list(range(16)), seed=42, acq=True), record_latents=True)
            per = {}
            for idx in sorted(refs):
                z, i = refs[idx]
                per[idx] = compare(res, z, idx, i)
            traces = {}
            for idx in sorted(refs):
                z, i = refs[idx]
                ti = list(z["trace_indices"])
                if idx in ti:
                    tr = torch.from_numpy(z["trace"][:, ti.index(idx)])  # [steps, 4, 64, 64]
                    traces[idx] = [(res["latents"][t][idx].float().cpu() - tr[t]).abs().max().item() for t in range(tr.shape[0])]
            worst = min(per, key=lambda k: per[k]["psnr_db"])
            report["ddim50"] = {"per_index": per, "per_step_latent_max_abs": traces, "indices": len(per),
                                "min_psnr_db": per[worst]["psnr_db"], "min_psnr_index": worst,
                                "mean_psnr_db": sum(v["psnr_db"] for v in per.values()) / len(per),
                                "max_latent_max_abs": max(v["latent_max_abs"] for v in per.values()),
                                "max_step_latent_max_abs": max((max(v) for v in traces.values()), default=None)}
            print(f"ddim50 [{args.tag}]: {len(per)} indices, PSNR min {per[worst]['psnr_db']:.2f} dB (index {worst}) mean "
                  f"{report['ddim50']['mean_psnr_db']:.2f} dB, latent max-abs {report['ddim50']['max_latent_max_abs']:.3e}", flush=True)
            print("   per index PSNR:", " ".join(f"{i}:{v['psnr_db']:.2f}" for i, v in per.items()), flush=True)
        elif case == "ddim200":
            out = {}
            for gs, name in ((0.0, "gs0"), (None, "gsNone")):
                path = os.path.join(GOLD, f"ddim200_{name}_sf001.npz")
                if not os.path.exists(path):
                    continue
                z = np.load(path)
                idx = [int(i) for i in z["indices"]]
                sch = ps.DDIMScheduler(clip_sample=False, **SCHED)
                sch.set_timesteps(200)
                s = Sampler(pc, pu, sch, pv, guidance_scale=gs, conditioning_scale=1.0, scale_factor=0.01)
                res = run(s, make_batch(idx, seed=42, acq=True))
                out[name] = {i: compare(res, z, k, k) for k, i in enumerate(idx)}
                print(f"ddim200 {name} [{args.tag}]:", " ".join(f"{i}: {v['psnr_db']:.2f} dB (latent {v['latent_max_abs']:.3e})"
                                                                  for i, v in out[name].items()), flush=True)
            if out:
                report["ddim200"] = out
        elif case == "ddpm_tail":
            path = os.path.join(GOLD, "ddpm_tail50_gs7.npz")
            if not os.path.exists(path):
                continue
            z = np.load(path)
            idx = [int(i) for i in z["indices"]]
            b = make_batch(idx, seed=42, acq=True)
            sch = ps.DDPMScheduler(**SCHED)
            assert sch.clip_sample is True
            sch.timesteps = torch.arange(49, -1, -1, dtype=torch.int64)
            start, dn = ddpm_tail_inputs(idx, sch)
            b["noise"] = start
            s = Sampler(pc, pu, sch, pv, guidance_scale=7.0, conditioning_scale=1.0, scale_factor=0.1)
            res = run(s, b, ddpm_noise=[n.to(dev) for n in dn], record_latents=True)
            tr = torch.from_numpy(z["trace"])  # [50, B, 4, 64, 64]
            steps = [(res["latents"][t].float().cpu() - tr[t]).abs().max().item() for t in range(tr.shape[0])]
            per = {i: compare(res, z, k, k) for k, i in enumerate(idx)}
            report["ddpm_tail"] = {"per_index": per, "per_step_latent_max_abs": steps, "max_step_latent_max_abs": max(steps)}
            print(f"ddpm_tail [{args.tag}]: per-step latent max-abs max {max(steps):.3e} (last {steps[-1]:.3e});",
                  " ".join(f"{i}: {v['psnr_db']:.2f} dB" for i, v in per.items()), flush=True)
    out = args.out or os.path.join(ROOT, "gpurun_out", f"g1_margin_{args.tag}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        json.dump(report, fh, indent=1)
    print(f"wrote {out} ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
