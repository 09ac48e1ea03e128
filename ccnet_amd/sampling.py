"""The ccnet sampling loop on the HIP path.

Mirrors, with the same names and argument meaning:
  * `sample_from_controlnet`  -- util.py:934-962
  * `sample_from_ldm`         -- util.py:1043-1092
  * the per-sample loop of run_controlnet_inference.py:211-256 (`sample`, `Sampler`)
  * `interval_mapping` -> uint8 of save_controlnet_inference_sample, util.py:1162-1173,1412-1416 (`to_uint8`)

`Sampler` is the MI355X-first form of that loop for a batch of B images: the latent is NOT duplicated for
classifier-free guidance (the edge conv reads image n % B), the step-invariant work is hoisted out of the N-step
loop (ControlNet conditioning embedding, every cross-attention K/V projection of the prompt embeddings,
SURVEY.md Appendix E), ControlNet residuals stay NHWC bf16 and are added inside the UNet kernels that produce the
skip tensors, and CFG + scheduler step are one kernel.  Results are the same computation the reference performs.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import ops
from .lib import CcnetError
from .schedulers import DDPMScheduler, _cfg_mode


def sample_from_controlnet(controlnet, noisy_e, controlnet_cond, timesteps, prompt_embeds, class_labels=None,
                           conditioning_scale=1.0):
    if class_labels is not None and len(class_labels.shape) == 2 and class_labels.shape[0] == 1 \
            and class_labels.shape[1] != 1:
        class_labels = torch.squeeze(class_labels, dim=0)  # (1, bs) -> (bs)
    if class_labels is not None and len(class_labels.shape) == 1:
        class_labels = torch.unsqueeze(class_labels, dim=1)  # (bs) -> (bs, 1)
    if hasattr(controlnet, "x") or "ControlNetModel" not in str(controlnet.__class__.__name__):
        return controlnet(x=noisy_e, timesteps=timesteps, context=prompt_embeds, controlnet_cond=controlnet_cond,
                          conditioning_scale=conditioning_scale, class_labels=class_labels)
    return controlnet(sample=noisy_e, timestep=timesteps, encoder_hidden_states=prompt_embeds,
                      controlnet_cond=controlnet_cond, conditioning_scale=conditioning_scale,
                      class_labels=class_labels, return_dict=False)  # diffusers dialect, util.py:953-961


def sample_from_ldm(model, noisy_e, timesteps, prompt_embeds, down_block_res_samples=None, mid_block_res_sample=None,
                    class_labels=None):
    if class_labels is not None and class_labels.shape[0] == 1 and class_labels.shape[0] != noisy_e.shape[0]:
        class_labels = torch.unsqueeze(torch.squeeze(class_labels, dim=0), dim=1)
    if hasattr(model, "x") or "DiffusionModelUNet" in str(model.__class__.__name__):
        if down_block_res_samples is None or mid_block_res_sample is None:
            return model(x=noisy_e, timesteps=timesteps, context=prompt_embeds, class_labels=class_labels)
        return model(x=noisy_e, timesteps=timesteps, context=prompt_embeds, class_labels=class_labels,
                     down_block_additional_residuals=down_block_res_samples,
                     mid_block_additional_residual=mid_block_res_sample)
    # diffusers dialect, util.py:1071-1091
    if down_block_res_samples is None or mid_block_res_sample is None:
        return model(sample=noisy_e, timestep=timesteps, encoder_hidden_states=prompt_embeds,
                     class_labels=class_labels).sample
    return model(sample=noisy_e, timestep=timesteps, encoder_hidden_states=prompt_embeds, class_labels=class_labels,
                 down_block_additional_residuals=down_block_res_samples,
                 mid_block_additional_residual=mid_block_res_sample).sample


def to_uint8(sample: torch.Tensor) -> torch.Tensor:
    """[-1,1] -> [0,1] clip -> rint(x255) -> uint8, channel 0 (util.py:1412-1416). Returns [B, H, W] uint8 on GPU."""
    return ops.to_uint8(sample)


class Sampler:
    """Batched ControlNet+UNet sampler with CFG; one instance per (controlnet, diffusion, scheduler)."""

    def __init__(self, controlnet, diffusion, scheduler, stage1=None, guidance_scale: Optional[float] = None,
                 conditioning_scale: float = 1.0, scale_factor: float = 0.01, cfg: bool = True,
                 use_graph: bool = True, overlap: Optional[bool] = None):
        # diffusers-dialect wrappers (ccnet_amd.diffusers_dialect) hold the lowered network as `_net`
        for m in (controlnet, diffusion):
            if hasattr(m, "_sync"):
                m._sync()
        controlnet, diffusion = getattr(controlnet, "_net", controlnet), getattr(diffusion, "_net", diffusion)
        self.controlnet, self.diffusion, self.scheduler, self.stage1 = controlnet, diffusion, scheduler, stage1
        self.guidance_scale = guidance_scale
        self.conditioning_scale = conditioning_scale
        self.scale_factor = scale_factor
        self.cfg = cfg  # run the [uncond, text] double batch as the reference does (run_controlnet_inference.py:219)
        # One denoising step is ~500 kernel launches; enqueued one by one from Python the host, not the GPU, sets the
        # pace.  With use_graph the step is captured once per input geometry into a HIP graph (static latent /
        # timestep / coefficient buffers) and replayed for every timestep.
        self.use_graph = use_graph
        self._graphs = {}
        # The UNet's encoder + middle block depend on (x, t) only (upstream adds the ControlNet residuals to the skip copies,
        # never to the trunk), so they run on a second stream beside the ControlNet; the ControlNet's zero-convolutions
        # then write skip + residual in their epilogue and the decoder continues on the main stream.  One image alone
        # leaves most CUs idle in either network; together they fill more of the chip.
        # Measured on one box (tools/ab_overlap.sh): +23 % at one image, +17 % at 2, +10 % at 4, +5 % at 8, +2.5 % at 16.
        # CCNET_OVERLAP=0 or overlap=False turns it off (profiling: overlapped kernels share CUs, so their individual
        # durations stop meaning anything -- it is also off while bench.py brackets launches with events).
        self.overlap = overlap if overlap is not None else os.environ.get("CCNET_OVERLAP", "1") != "0"
        self._side = None
        # guidance_scale None keeps the text half of the [uncond, text] batch and 0 the unconditional half
        # (run_controlnet_inference.py:245-252; the reference's canonical script runs guidance_scale=0.0,
        # src/bash/INF_CTRL_CCNET_P1.sh:21).  No operation couples the rows of a batch, so only the half that is kept is
        # evaluated: B rows instead of 2B, the same value per image up to the order of fp32 sums (tile plans depend on
        # the row count).  CCNET_CFG_PRUNE=0 evaluates both halves and discards one, as the reference does.
        self.prune_cfg = os.environ.get("CCNET_CFG_PRUNE", "1") != "0"

    def _one_branch(self) -> bool:
        return self.cfg and self.prune_cfg and (self.guidance_scale is None or self.guidance_scale == 0)

    @torch.no_grad()
    def prepare(self, cond: torch.Tensor, prompt_embeds: torch.Tensor, empty_prompt_embeds: torch.Tensor,
                acquisition_time: Optional[torch.Tensor] = None):
        """Step-invariant state for a batch: cond (B,1,H,W) in [-1,1]; prompt_embeds (B,77,D);
        empty_prompt_embeds (1,77,D); acquisition_time (B,) seconds or None."""
        if not cond.is_cuda:
            raise CcnetError("Sampler inputs must be on the GPU (no CPU path)")
        B = cond.shape[0]
        two = self.cfg and not self._one_branch()
        if two:
            # [uncond x B, text x B]: empty prompt first (run_controlnet_inference.py:207-208)
            ctx = torch.cat([empty_prompt_embeds.to(prompt_embeds).expand(B, -1, -1), prompt_embeds], dim=0)
        elif self.cfg and self.guidance_scale is not None:  # guidance 0: the unconditional half alone
            ctx = empty_prompt_embeds.to(prompt_embeds).expand(B, -1, -1)
        else:
            ctx = prompt_embeds
        st = {"B": B, "Bc": ctx.shape[0]}
        st["kv_c"] = self.controlnet.context_kv(ctx.contiguous())
        st["kv_u"] = self.diffusion.context_kv(ctx.contiguous())
        st["cond_emb"] = self.controlnet.cond_embedding(cond)  # B images; image n of the CFG batch reads n % B
        cls = None
        if acquisition_time is not None:
            a = acquisition_time.reshape(B).float()
            cls = (torch.cat([a, a]) if two else a).reshape(-1, 1).to(cond.device)
        st["cls"] = cls
        # every step's time-embedding projections in one batched evaluation per network ([T, Bc, channels]; None if too big)
        tables = os.environ.get("CCNET_EMB_TABLE", "1") != "0"  # (switch for A/B timing)
        st["tall_c"] = self.controlnet.embedding_table(self.scheduler.timesteps, cls, st["Bc"]) if tables else None
        st["tall_u"] = self.diffusion.embedding_table(self.scheduler.timesteps, cls, st["Bc"]) if tables else None
        return st

    def _model_output(self, st, x: torch.Tensor, ts: torch.Tensor, tall_c=None, tall_u=None) -> torch.Tensor:
        """ControlNet + UNet on the latent x at timestep tensor ts -> model output fp32 [Bc, C, h, w].  tall_c / tall_u:
        this step's rows of the embedding tables of prepare() (None: the networks evaluate their embedding MLPs)."""
        # [uncond x B, text x B] from ONE latent batch: image n and n + B share latent, conditioning image, timestep and
        # acquisition time (prepare() builds cls as [a, a]) until the first cross-attention reads their prompts
        shared = self.cfg and st["Bc"] == 2 * x.shape[0]
        if not self.overlap or ops.prof_is_enabled():
            down, mid = self.controlnet.forward_native(
                x, ts, None, self.conditioning_scale, class_labels=st["cls"], _ctx_kv=st["kv_c"],
                _cond_emb=st["cond_emb"], _x_batch=st["Bc"], _tall=tall_c, _cfg_shared=shared)
            return self.diffusion(x, ts, class_labels=st["cls"], down_block_additional_residuals=down,
                                  mid_block_additional_residual=mid, _ctx_kv=st["kv_u"], _x_batch=st["Bc"], _tall=tall_u,
                                  _cfg_shared=shared)
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=x.device)
        side = self._side
        side.wait_stream(cur)  # fork: everything the side stream touches was produced on `cur` (also inside a graph capture)
        with torch.cuda.stream(side):
            enc = self.diffusion.encode_phase(x, ts, class_labels=st["cls"], _ctx_kv=st["kv_u"], _x_batch=st["Bc"],
                                              _tall=tall_u, _cfg_shared=shared)
        skips, mid = self.controlnet.forward_native(
            x, ts, None, self.conditioning_scale, class_labels=st["cls"], _ctx_kv=st["kv_c"],
            _cond_emb=st["cond_emb"], _x_batch=st["Bc"], _unet_state=enc, _before_zero=lambda: cur.wait_stream(side),
            _tall=tall_c, _cfg_shared=shared)
        return self.diffusion.decode_phase(enc, skips, mid)

    @torch.no_grad()
    def step(self, st, x: torch.Tensor, t, ddpm_noise: Optional[torch.Tensor] = None, index: Optional[int] = None):
        """One denoising step on the fp32 latent x (B,C,h,w); returns (x_prev, x0).  `index`: position of t in
        scheduler.timesteps (selects the rows of prepare()'s embedding tables; looked up when omitted)."""
        ts = torch.tensor([float(t)], dtype=torch.float32, device=x.device)
        tall_c = tall_u = None
        if st.get("tall_c") is not None and st.get("tall_u") is not None:
            if index is None:
                index = [int(v) for v in self.scheduler.timesteps].index(int(t))
            tall_c, tall_u = st["tall_c"][index], st["tall_u"][index]
        out = self._model_output(st, x, ts, tall_c, tall_u)
        if self.cfg and not self._one_branch():
            if isinstance(self.scheduler, DDPMScheduler):
                return self.scheduler.step_cfg(out, t, x, self.guidance_scale, noise=ddpm_noise)
            return self.scheduler.step_cfg(out, t, x, self.guidance_scale)
        if isinstance(self.scheduler, DDPMScheduler):
            return self.scheduler.step(out, t, x, noise=ddpm_noise)
        return self.scheduler.step(out, t, x)

    # ------------------------------------------------------------------ HIP-graph replay of the denoising step
    def _static_step(self, g):
        """The step on static buffers: ControlNet -> UNet -> fused CFG + scheduler update of g["x"] in place."""
        st = g["st"]
        out = self._model_output(st, g["x"], g["ts"], g["tall_c"], g["tall_u"])
        if self.cfg and not self._one_branch():
            mode, gs = _cfg_mode(self.guidance_scale)
        else:
            mode, gs = 3, 0.0  # model_out is the prediction itself
        ops.cfg_scheduler_step(out, mode, gs, g["coef"], g["x"], noise=g["noise"])

    @staticmethod
    def _copy_state(dst, src):
        """Copy the per-batch step-invariant tensors of `src` into the static buffers the graph was captured on."""
        for k in ("kv_c", "kv_u"):
            for d, s_ in zip(dst[k][0], src[k][0]):
                d.copy_(s_)
                if getattr(d, "_ximg", None) is not None:  # (the same keys / values laid out for the fused cross-attention kernel)
                    d._ximg.copy_(s_._ximg)
        dst["cond_emb"].t.copy_(src["cond_emb"].t)
        if src["cls"] is not None:
            dst["cls"].copy_(src["cls"])

    def _graph_for(self, st, x: torch.Tensor, ddpm: bool):
        # Everything the capture bakes into the graph is part of the key: kernel scalars (guidance / conditioning scale, CFG
        # mode, the scheduler's prediction type and clipping live in `coef`, refreshed per step), the launch topology
        # (cfg, overlap) and the device addresses of the packed weights.  A network re-packs after load_state_dict / .to() /
        # adjust_class_embedding (new tensors, the old ones are freed): its pack generation then changes and every
        # cached graph is dropped, so no replay can read freed weights.
        gen = (getattr(self.controlnet, "_pk_gen", 0), getattr(self.diffusion, "_pk_gen", 0))
        if gen != getattr(self, "_graph_gen", None):
            self._graphs.clear()
            self._graph_gen = gen
        key = (tuple(x.shape), st["Bc"], st["cls"] is None, st["kv_c"][1], ddpm, id(self.scheduler),
               tuple(int(t) for t in self.scheduler.timesteps[:2]), len(self.scheduler.timesteps),
               None if self.guidance_scale is None else float(self.guidance_scale), float(self.conditioning_scale),
               bool(self.cfg), bool(self._one_branch()), bool(self.overlap), st.get("tall_c") is not None,
               getattr(self.scheduler, "clip_sample", None), getattr(self.scheduler, "prediction_type", None))
        g = self._graphs.get(key)
        if g is not None:
            self._copy_state(g["st"], st)
            return g
        dev = x.device
        ts_host = [float(t) for t in self.scheduler.timesteps]
        g = {"st": st, "x": x.clone(), "ts": torch.zeros(1, dtype=torch.float32, device=dev),
             "coef": torch.zeros(8, dtype=torch.float32, device=dev),
             "noise": torch.zeros_like(x) if ddpm else None,
             "ts_table": torch.tensor(ts_host, dtype=torch.float32, device=dev).reshape(-1, 1),
             "coef_table": torch.stack([self.scheduler.coef_tensor(int(t), dev) for t in self.scheduler.timesteps]),
             "pk": (self.controlnet._pk, self.diffusion._pk)}  # the captured weight pointers stay alive with the graph
        tables = st.get("tall_c") is not None and st.get("tall_u") is not None
        g["tall_c"] = st["tall_c"][0].clone() if tables else None  # this step's embedding rows (static buffers)
        g["tall_u"] = st["tall_u"][0].clone() if tables else None
        g["ts"].copy_(g["ts_table"][0])
        g["coef"].copy_(g["coef_table"][0])
        self._static_step(g)  # eager once: weight packing, kernel attributes and allocator warm-up stay out of the graph
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of other threads (the RCCL watchdog of a multi-GPU run polls events) must not
        # invalidate this thread's capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._static_step(g)
        g["graph"] = graph
        self._graphs[key] = g
        return g

    def _sample_graphed(self, st, noise, ddpm_noise, record_latents, max_steps):
        ddpm = isinstance(self.scheduler, DDPMScheduler)
        x0 = noise.float().contiguous()
        g = self._graph_for(st, x0, ddpm)
        g["x"].copy_(x0)
        trace = []
        for i, t in enumerate(self.scheduler.timesteps):
            if max_steps is not None and i >= max_steps:
                break
            g["ts"].copy_(g["ts_table"][i])
            g["coef"].copy_(g["coef_table"][i])
            if g["tall_c"] is not None:
                g["tall_c"].copy_(st["tall_c"][i])
                g["tall_u"].copy_(st["tall_u"][i])
            if ddpm:
                if int(t) > 0:
                    n = ddpm_noise[i] if ddpm_noise is not None else torch.randn(x0.shape, dtype=torch.float32)
                    g["noise"].copy_(n.to(x0.device, non_blocking=True))
                else:
                    g["noise"].zero_()  # sigma is 0 at t = 0: upstream adds no noise there
            g["graph"].replay()
            if record_latents:
                trace.append(g["x"].clone())
        return g["x"].clone(), trace

    @torch.no_grad()
    def sample(self, noise: torch.Tensor, cond: torch.Tensor, prompt_embeds: torch.Tensor,
               empty_prompt_embeds: torch.Tensor, acquisition_time: Optional[torch.Tensor] = None,
               ddpm_noise: Optional[List[torch.Tensor]] = None, record_latents: bool = False,
               max_steps: Optional[int] = None, decode: bool = True):
        st = self.prepare(cond, prompt_embeds, empty_prompt_embeds, acquisition_time)
        if self.use_graph and not ops.prof_is_enabled():
            x, trace = self._sample_graphed(st, noise, ddpm_noise, record_latents, max_steps)
        else:
            x = noise.float().contiguous()
            trace = []
            for i, t in enumerate(self.scheduler.timesteps):
                if max_steps is not None and i >= max_steps:
                    break
                x, _ = self.step(st, x, int(t), None if ddpm_noise is None else ddpm_noise[i], index=i)
                if record_latents:
                    trace.append(x.clone())
        res = {"latent": x}
        if record_latents:
            res["latents"] = trace
        if decode and self.stage1 is not None:
            # run_controlnet_inference.py:255-256
            img = self.stage1.decode(x / self.scale_factor)
            # diffusers' AutoencoderKL returns DecoderOutput; the reference unwraps `.sample` when saving (util.py:1402)
            res["image"] = getattr(img, "sample", img)
        return res


def sample(controlnet, diffusion, scheduler, noise, cond, prompt_embeds, empty_prompt_embeds,
           guidance_scale=None, conditioning_scale=1.0, acquisition_time=None, stage1=None, scale_factor=0.01,
           **kw):
    """Functional form of the reference loop for one batch (run_controlnet_inference.py:211-256)."""
    s = Sampler(controlnet, diffusion, scheduler, stage1, guidance_scale, conditioning_scale, scale_factor)
    return s.sample(noise, cond, prompt_embeds, empty_prompt_embeds, acquisition_time, **kw)


@torch.no_grad()
def encode_prompts(text_encoder, input_ids: torch.Tensor) -> torch.Tensor:
    """`text_encoder(reports.squeeze(1))[0]` (run_controlnet_inference.py:149-150,205): token ids [B,77] or [B,1,77]
    -> prompt embeddings [B,77,D] on the text tower's device."""
    ids = input_ids.reshape(-1, input_ids.shape[-1])
    return text_encoder(ids.to(text_encoder.device))[0]


@torch.no_grad()
def preview_controlnet_sample(cond, controlnet, diffusion, stage1, text_encoder, scheduler, spatial_shape=None,
                              device=None, scale_factor: float = 1.0, acquisition_time=None, text_condition=None,
                              controlnet_conditioning_scale: float = 1.0, noise: Optional[torch.Tensor] = None):
    """The sampling part of `log_controlnet_samples` (util.py:840-931): one image, no CFG, prompt through the text tower
    (BOS + 76 x EOS ids when `text_condition` is None, :873), every timestep of `scheduler`, AEKL decode of
    latent / scale_factor (:918).  Returns dict(image, latent, prompt_embeds); the figure / TensorBoard part of the
    reference function stays with the caller (`ccnet_amd.io.save_controlnet_inference_sample` writes the PNGs)."""
    device = torch.device(device) if device is not None else cond.device
    if noise is None:
        noise = torch.randn((1,) + tuple(spatial_shape))  # :869, CPU generator as upstream
    if text_condition is None:
        ids = torch.cat((49406 * torch.ones(1, 1), 49407 * torch.ones(1, 76)), 1).long()
    else:
        ids = text_condition
    prompt_embeds = encode_prompts(text_encoder, ids.to(device))
    s = Sampler(controlnet, diffusion, scheduler, getattr(stage1, "model", stage1), guidance_scale=None,
                conditioning_scale=controlnet_conditioning_scale, scale_factor=scale_factor, cfg=False)
    acq = None if acquisition_time is None else acquisition_time.reshape(-1).to(device)
    res = s.sample(noise.to(device), cond.to(device), prompt_embeds, prompt_embeds[:1], acq)
    res["prompt_embeds"] = prompt_embeds
    return res
