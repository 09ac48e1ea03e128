"""fp32 restatement of monai-generative's DDIM / DDPM schedulers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).

Reference call sites: run_controlnet_inference.py:100-112 (construction),
:216 (`.timesteps`), :253 (`.step(noise_pred, t, noise) -> (prev, x0)`),
training_functions.py:931,960 (`add_noise`, `get_velocity`).
Known-answer constants that pin this file: SURVEY.md Appendix C.
"""
from __future__ import annotations

import numpy as np
import torch


class Scheduler:
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", **schedule_args):
        self.num_train_timesteps = num_train_timesteps
        b0 = float(schedule_args.get("beta_start", 1e-4))
        b1 = float(schedule_args.get("beta_end", 2e-2))
        if schedule == "linear_beta":
            self.betas = torch.linspace(b0, b1, num_train_timesteps, dtype=torch.float32)
        elif schedule == "scaled_linear_beta":
            self.betas = torch.linspace(b0 ** 0.5, b1 ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"schedule {schedule}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].astype(np.int64))

    @staticmethod
    def _expand(v, like):
        v = v.flatten()
        while v.dim() < like.dim():
            v = v.unsqueeze(-1)
        return v

    def add_noise(self, original_samples, noise, timesteps):
        acp = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        a = self._expand(acp[timesteps] ** 0.5, original_samples)
        s = self._expand((1 - acp[timesteps]) ** 0.5, original_samples)
        return a * original_samples + s * noise

    def get_velocity(self, sample, noise, timesteps):
        acp = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)
        timesteps = timesteps.to(sample.device)
        a = self._expand(acp[timesteps] ** 0.5, sample)
        s = self._expand((1 - acp[timesteps]) ** 0.5, sample)
        return a * noise - s * sample

    def _x0_eps(self, prediction_type, model_output, sample, alpha_prod_t):
        beta_prod_t = 1 - alpha_prod_t
        if prediction_type == "epsilon":
            x0 = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
            eps = model_output
        elif prediction_type == "sample":
            x0 = model_output
            eps = (sample - alpha_prod_t ** 0.5 * x0) / beta_prod_t ** 0.5
        elif prediction_type == "v_prediction":
            x0 = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output
            eps = (alpha_prod_t ** 0.5) * model_output + (beta_prod_t ** 0.5) * sample
        else:
            raise ValueError(prediction_type)
        return x0, eps


class DDIMScheduler(Scheduler):
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta", clip_sample: bool = True,
                 set_alpha_to_one: bool = True, steps_offset: int = 0, prediction_type: str = "epsilon",
                 **schedule_args):
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        self.prediction_type = prediction_type
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.clip_sample = clip_sample
        self.steps_offset = steps_offset
        self.set_timesteps(num_train_timesteps)

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device) + self.steps_offset

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None):
        t = int(timestep)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0, eps = self._x0_eps(self.prediction_type, model_output, sample, a_t)
        if self.clip_sample:
            x0 = torch.clamp(x0, -1, 1)
        variance = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
        std = eta * variance ** 0.5
        direction = (1 - a_p - std ** 2) ** 0.5 * eps
        prev_sample = a_p ** 0.5 * x0 + direction
        if eta > 0:
            noise = torch.randn(model_output.shape, dtype=model_output.dtype, generator=generator).to(model_output.device)
            prev_sample = prev_sample + std * noise
        return prev_sample, x0


class DDPMScheduler(Scheduler):
    def __init__(self, num_train_timesteps: int = 1000, schedule: str = "linear_beta",
                 variance_type: str = "fixed_small", clip_sample: bool = True, prediction_type: str = "epsilon",
                 **schedule_args):
        super().__init__(num_train_timesteps, schedule, **schedule_args)
        if variance_type != "fixed_small":
            raise NotImplementedError("reference config only uses the default fixed_small variance")
        self.variance_type = variance_type
        self.clip_sample = clip_sample
        self.prediction_type = prediction_type

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)

    def step(self, model_output, timestep, sample, generator=None, noise=None):
        """`noise` (optional) injects the ancestral noise explicitly so parity tests need no shared RNG."""
        t = int(timestep)
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[t - 1] if t > 0 else self.one
        x0, _ = self._x0_eps(self.prediction_type, model_output, sample, a_t)
        if self.clip_sample:
            x0 = torch.clamp(x0, -1, 1)
        c_x0 = (a_p ** 0.5 * self.betas[t]) / (1 - a_t)
        c_xt = self.alphas[t] ** 0.5 * (1 - a_p) / (1 - a_t)
        prev_sample = c_x0 * x0 + c_xt * sample
        if t > 0:
            if noise is None:
                noise = torch.randn(model_output.size(), dtype=model_output.dtype, generator=generator)
            var = torch.clamp((1 - a_p) / (1 - a_t) * self.betas[t], min=1e-20)
            prev_sample = prev_sample + (var ** 0.5) * noise.to(model_output.device)
        return prev_sample, x0
