"""fp32 restatement of the reference's sampling loop and its two dispatch helpers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).

Follows run_controlnet_inference.py:211-256 (loop, CFG combine :245-252,
scheduler.step :253, decode :255-256), util.py:934-962 (sample_from_controlnet),
util.py:1043-1092 (sample_from_ldm) and util.py:1162-1173,1412-1416
(interval_mapping -> uint8).  The reference loop is batch-1 ([empty, text]
ordering); this restatement keeps that ordering
generalised to [uncond x B, text x B] and takes the initial noise explicitly
instead of drawing it from the global CPU generator (:212).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


def sample_from_controlnet(controlnet, noisy_e, controlnet_cond, timesteps, prompt_embeds, class_labels=None,
                           conditioning_scale=1.0):
    # util.py:936-940
    if class_labels is not None and class_labels.dim() == 2 and class_labels.shape[0] == 1 and class_labels.shape[1] != 1:
        class_labels = torch.squeeze(class_labels, dim=0)
    if class_labels is not None and class_labels.dim() == 1:
        class_labels = torch.unsqueeze(class_labels, dim=1)
    return controlnet(x=noisy_e, timesteps=timesteps, context=prompt_embeds, controlnet_cond=controlnet_cond,
                      conditioning_scale=conditioning_scale, class_labels=class_labels)


def sample_from_ldm(model, noisy_e, timesteps, prompt_embeds, down_block_res_samples=None,
                    mid_block_res_sample=None, class_labels=None):
    # util.py:1045-1048
    if class_labels is not None and class_labels.shape[0] == 1 and class_labels.shape[0] != noisy_e.shape[0]:
        class_labels = torch.unsqueeze(torch.squeeze(class_labels, dim=0), dim=1)
    if down_block_res_samples is None or mid_block_res_sample is None:
        return model(x=noisy_e, timesteps=timesteps, context=prompt_embeds, class_labels=class_labels)
    return model(x=noisy_e, timesteps=timesteps, context=prompt_embeds, class_labels=class_labels,
                 down_block_additional_residuals=down_block_res_samples,
                 mid_block_additional_residual=mid_block_res_sample)


def cfg_combine(model_output, guidance_scale):
    """run_controlnet_inference.py:245-252."""
    uncond, text = model_output.chunk(2)
    if guidance_scale is not None:
        if guidance_scale == 0:
            return uncond
        return uncond + guidance_scale * (text - uncond)
    return text


@torch.no_grad()
def sample(controlnet, diffusion, scheduler, noise, cond, prompt_embeds, empty_prompt_embeds,
           guidance_scale: Optional[float] = 7.0, conditioning_scale: float = 1.0, acquisition_time=None,
           stage1=None, scale_factor: float = 0.1, ddpm_noise=None, record_latents: bool = False,
           max_steps: Optional[int] = None):
    """noise (B,C,h,w); cond (B,1,H,W); prompt_embeds (B,77,D); empty_prompt_embeds (1,77,D);
    acquisition_time (B,) or None.  Returns dict(latent, image?, latents?)."""
    B = noise.shape[0]
    ctx = torch.cat([empty_prompt_embeds.expand(B, -1, -1), prompt_embeds], dim=0)
    cls = None
    if acquisition_time is not None:
        cls = torch.cat([acquisition_time.reshape(B), acquisition_time.reshape(B)]).reshape(2 * B, 1).float()
    x = noise.clone()
    trace = []
    for i, t in enumerate(scheduler.timesteps):
        if max_steps is not None and i >= max_steps:
            break
        timesteps = torch.Tensor((t,)).to(x.device).long()
        x_in = torch.cat([x] * 2)
        c_in = torch.cat([cond] * 2)
        down, mid = sample_from_controlnet(controlnet, x_in, c_in, timesteps, ctx, cls, conditioning_scale)
        out = sample_from_ldm(diffusion, x_in, timesteps, ctx, down, mid, cls)
        pred = cfg_combine(out, guidance_scale)
        if ddpm_noise is not None:
            x, _ = scheduler.step(pred, t, x, noise=ddpm_noise[i])
        else:
            x, _ = scheduler.step(pred, t, x)
        if record_latents:
            trace.append(x.clone())
    res = {"latent": x}
    if record_latents:
        res["latents"] = trace
    if stage1 is not None:
        res["image"] = stage1.decode(x / scale_factor)
    return res


@torch.no_grad()
def preview_sample(cond, controlnet, diffusion, stage1, text_forward, scheduler, noise, scale_factor: float = 1.0,
                   acquisition_time=None, text_condition=None, conditioning_scale: float = 1.0):
    """util.py:840-931 (log_controlnet_samples) without the figure: one image, no CFG, the prompt run through the text
    tower (`text_forward(ids) -> hidden states`); with no prompt the ids are BOS followed by 76 EOS (:873).  The initial
    latent is passed in instead of drawn at :869.  Returns dict(latent, image, prompt_embeds)."""
    if text_condition is None:
        ids = torch.cat((49406 * torch.ones(1, 1), 49407 * torch.ones(1, 76)), 1).long()
    else:
        ids = text_condition
    prompt_embeds = text_forward(ids.squeeze(1))
    latent = noise.clone()
    for t in scheduler.timesteps:
        timestep = torch.asarray((t,))
        down, mid = sample_from_controlnet(controlnet, latent, cond, timestep, prompt_embeds, acquisition_time,
                                           conditioning_scale)
        noise_pred = sample_from_ldm(diffusion, latent, timestep, prompt_embeds, down, mid, acquisition_time)
        latent, _ = scheduler.step(noise_pred, t, latent)
    return {"latent": latent, "image": stage1.decode(latent / scale_factor), "prompt_embeds": prompt_embeds}


def interval_mapping(image, from_min, from_max, to_min, to_max, is_clipped=True):
    """util.py:1162-1173."""
    scaled = np.array((image - from_min) / float(from_max - from_min), dtype=float)
    x = to_min + scaled * (to_max - to_min)
    return np.clip(x, to_min, to_max) if is_clipped else x


def to_uint8(sample: torch.Tensor) -> np.ndarray:
    """util.py:1405-1416: [-1,1] -> [0,1] clip -> rint(x255) -> uint8, channel 0 of each image."""
    a = sample.detach().cpu().numpy().astype(np.float64)
    a = interval_mapping(a, -1.0, 1.0, 0.0, 1.0)
    return np.rint(a * 255).astype(np.uint8)[:, 0]
