"""The diffusers dialect of the two denoising networks: `UNet2DConditionModel` and `ControlNetModel`.

The reference's SD-2.1 path builds these diffusers classes (`util.py:1232`, `util.py:1342`
`ControlNetModel.from_unet`), loads `ckpt["diffusion"]` / `ckpt["controlnet"]` state dicts in diffusers key names
(`util.py:1305,1364`) and calls them with diffusers kwargs (`sample=, timestep=, encoder_hidden_states=, ...`,
`util.py:953-961,1084-1091`); dispatch is by class name (`"ControlNetModel" in cls.__name__`, `util.py:943,1051`).

The classes here keep that surface -- constructor kwargs, `.config`, `forward` kwargs and return types,
`state_dict()` / `load_state_dict()` in diffusers key names, `ControlNetModel.from_unet`, and a LOCAL-directory
`from_pretrained` (config.json + safetensors / .bin; there is no hub access) -- over the same MI355X lowering as
`ccnet_amd.nets` (the two libraries implement the same network; what differs is naming, the Linear-vs-1x1-conv
spelling of `proj_in` / `proj_out` under `use_linear_projection`, and the Transformer2DModel GroupNorm eps, fixed at
1e-6 in diffusers).  Options of the diffusers classes that the SD-2.1 family does not use raise NotImplementedError.
"""
from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import nets
from .lib import CcnetError


class FrozenConfig(dict):
    """Attribute + item access, like diffusers' FrozenDict -- but assignable, because the reference assigns
    `model.config.num_class_embeds` (util.py:1114,1337,1350)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor


@dataclass
class ControlNetOutput:
    down_block_res_samples: Tuple[torch.Tensor, ...]
    mid_block_res_sample: torch.Tensor


# ------------------------------------------------------------------------------------------------ key translation
# (diffusers pattern, monai-generative replacement), applied in order; the inverse list below mirrors it
_D2M = [
    (r"^conv_in\.(weight|bias)$", r"conv_in.conv.\1"),
    (r"^time_embedding\.linear_1\.", "time_embed.0."),
    (r"^time_embedding\.linear_2\.", "time_embed.2."),
    (r"^mid_block\.resnets\.0\.", "middle_block.resnet_1."),
    (r"^mid_block\.resnets\.1\.", "middle_block.resnet_2."),
    (r"^mid_block\.attentions\.0\.", "middle_block.attention."),
    (r"^(down_blocks\.\d+)\.downsamplers\.0\.conv\.", r"\1.downsampler.op.conv."),
    (r"^(up_blocks\.\d+)\.upsamplers\.0\.conv\.", r"\1.upsampler.conv.conv."),
    (r"^conv_norm_out\.", "out.0."),
    (r"^conv_out\.", "out.2.conv."),
    (r"^controlnet_cond_embedding\.conv_in\.", "controlnet_cond_embedding.conv_in.conv."),
    (r"^controlnet_cond_embedding\.conv_out\.", "controlnet_cond_embedding.conv_out.conv."),
    (r"^(controlnet_cond_embedding\.blocks\.\d+)\.", r"\1.conv."),
    (r"\.conv_shortcut\.", ".skip_connection.conv."),
    (r"\.(conv1|conv2)\.(weight|bias)$", r".\1.conv.\2"),
    (r"\.(proj_in|proj_out)\.(weight|bias)$", r".\1.conv.\2"),
    (r"\.ff\.net\.0\.proj\.", ".ff.linear1."),
    (r"\.ff\.net\.2\.", ".ff.linear2."),
]
_M2D = [
    (r"^conv_in\.conv\.", "conv_in."),
    (r"^time_embed\.0\.", "time_embedding.linear_1."),
    (r"^time_embed\.2\.", "time_embedding.linear_2."),
    (r"^middle_block\.resnet_1\.", "mid_block.resnets.0."),
    (r"^middle_block\.resnet_2\.", "mid_block.resnets.1."),
    (r"^middle_block\.attention\.", "mid_block.attentions.0."),
    (r"^(down_blocks\.\d+)\.downsampler\.op\.conv\.", r"\1.downsamplers.0.conv."),
    (r"^(up_blocks\.\d+)\.upsampler\.conv\.conv\.", r"\1.upsamplers.0.conv."),
    (r"^out\.0\.", "conv_norm_out."),
    (r"^out\.2\.conv\.", "conv_out."),
    (r"^controlnet_cond_embedding\.conv_in\.conv\.", "controlnet_cond_embedding.conv_in."),
    (r"^controlnet_cond_embedding\.conv_out\.conv\.", "controlnet_cond_embedding.conv_out."),
    (r"^(controlnet_cond_embedding\.blocks\.\d+)\.conv\.", r"\1."),
    (r"\.skip_connection\.conv\.", ".conv_shortcut."),
    (r"\.(conv1|conv2)\.conv\.", r".\1."),
    (r"\.(proj_in|proj_out)\.conv\.", r".\1."),
    (r"\.ff\.linear1\.", ".ff.net.0.proj."),
    (r"\.ff\.linear2\.", ".ff.net.2."),
]


def _translate(key: str, table) -> str:
    for pat, rep in table:
        key = re.sub(pat, rep, key)
    return key


def diffusers_key_to_monai(key: str) -> str:
    return _translate(key, _D2M)


def monai_key_to_diffusers(key: str) -> str:
    return _translate(key, _M2D)


_PROJ = re.compile(r"\.(proj_in|proj_out)\.(conv\.)?weight$")


def state_dict_to_monai(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """diffusers-named state dict -> the names `ccnet_amd.nets` (monai-generative) uses; Linear `proj_in`/`proj_out`
    weights ([C, C], `use_linear_projection=True`) become 1x1 conv weights ([C, C, 1, 1])."""
    out = {}
    for k, v in sd.items():
        mk = diffusers_key_to_monai(k)
        if _PROJ.search(mk) and v.dim() == 2:
            v = v[:, :, None, None]
        out[mk] = v
    return out


def state_dict_to_diffusers(sd: Dict[str, torch.Tensor], use_linear_projection: bool) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        dk = monai_key_to_diffusers(k)
        if use_linear_projection and _PROJ.search(dk) and v.dim() == 4:
            v = v[:, :, 0, 0]
        out[dk] = v
    return out


# ------------------------------------------------------------------------------------------------ shared wrapper
def _check_unused(**kw):
    bad = [k for k, v in kw.items() if v is not None and v is not False]
    if bad:
        raise NotImplementedError(f"diffusers option(s) {bad} are not used by the SD-2.1 family and are not lowered")


def _geometry(cfg) -> dict:
    """diffusers config -> monai-generative kwargs of the same network."""
    boc = list(cfg["block_out_channels"])
    n = len(boc)
    dbt = list(cfg["down_block_types"])
    if len(dbt) != n or any(t not in ("CrossAttnDownBlock2D", "DownBlock2D") for t in dbt):
        raise NotImplementedError(f"down_block_types {dbt}")
    att = [t == "CrossAttnDownBlock2D" for t in dbt]
    # diffusers passes `attention_head_dim` as the NUMBER of heads when num_attention_heads is None
    # (UNet2DConditionModel.__init__: `num_attention_heads = num_attention_heads or attention_head_dim`)
    heads = cfg.get("num_attention_heads") or cfg["attention_head_dim"]
    heads = list(heads) if isinstance(heads, (list, tuple)) else [heads] * n
    head_ch = [c // h for c, h in zip(boc, heads)]
    lpb = cfg["layers_per_block"]
    if isinstance(lpb, (list, tuple)):
        raise NotImplementedError("per-block layers_per_block")
    if cfg.get("act_fn", "silu") not in ("silu", "swish"):
        raise NotImplementedError(f"act_fn {cfg['act_fn']}")
    if not cfg.get("flip_sin_to_cos", True) or cfg.get("freq_shift", 0) != 0:
        raise NotImplementedError("flip_sin_to_cos=False / freq_shift != 0 timestep embeddings")
    tl = cfg.get("transformer_layers_per_block", 1)
    if tl not in (1, [1] * n, (1,) * n):
        raise NotImplementedError("transformer_layers_per_block != 1")
    ted = cfg.get("time_embedding_dim")
    if ted not in (None, boc[0] * 4):
        raise NotImplementedError("time_embedding_dim != 4 * block_out_channels[0]")
    _check_unused(class_embed_type=cfg.get("class_embed_type"), addition_embed_type=cfg.get("addition_embed_type"),
                  encoder_hid_dim=cfg.get("encoder_hid_dim"), dual_cross_attention=cfg.get("dual_cross_attention"),
                  only_cross_attention=cfg.get("only_cross_attention"), upcast_attention=cfg.get("upcast_attention"),
                  time_cond_proj_dim=cfg.get("time_cond_proj_dim"), timestep_post_act=cfg.get("timestep_post_act"),
                  conv_in_kernel=None if cfg.get("conv_in_kernel", 3) == 3 else cfg.get("conv_in_kernel"),
                  resnet_time_scale_shift=None if cfg.get("resnet_time_scale_shift", "default") == "default" else 1)
    cad = cfg["cross_attention_dim"]
    if isinstance(cad, (list, tuple)):
        raise NotImplementedError("per-block cross_attention_dim")
    return dict(spatial_dims=2, in_channels=cfg["in_channels"], num_res_blocks=lpb, num_channels=boc,
                attention_levels=att, norm_num_groups=cfg.get("norm_num_groups", 32) or 32,
                norm_eps=cfg.get("norm_eps", 1e-5), num_head_channels=head_ch, with_conditioning=True,
                cross_attention_dim=cad, num_class_embeds=cfg.get("num_class_embeds"))


class _DiffusersNet(nn.Module):
    """Holds the lowered network as `_net` and speaks the diffusers state-dict / attribute dialect around it."""

    config: FrozenConfig

    def _setup(self, net: nn.Module, config: dict):
        object.__setattr__(self, "config", FrozenConfig(config))
        self._net = net
        net.transformer_norm_eps = 1e-6  # Transformer2DModel.norm (diffusers) is GroupNorm(eps=1e-6)

    # `model.class_embedding = nn.Sequential(...)` (util.py:1125) must land in the lowered network
    def __setattr__(self, name, value):
        if name == "class_embedding" and "_net" in self._modules:
            self._net.class_embedding = value
            self._net._invalidate()
            return
        super().__setattr__(name, value)

    def __getattr__(self, name):
        if name == "class_embedding":
            return getattr(self._modules["_net"], "class_embedding")
        return super().__getattr__(name)

    def _sync(self):
        n = self.config.get("num_class_embeds")
        if self._net.num_class_embeds != n:
            self._net.num_class_embeds = n
            self._net._invalidate()

    @property
    def device(self):
        return next(self._net.parameters()).device

    @property
    def dtype(self):
        return torch.float32

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = self._net.state_dict(keep_vars=keep_vars)
        out = destination if destination is not None else {}
        for k, v in state_dict_to_diffusers(sd, bool(self.config.get("use_linear_projection"))).items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        return self._net.load_state_dict(state_dict_to_monai(dict(state_dict)), strict=strict, **kw)

    @classmethod
    def from_config(cls, config: dict, **kw):
        c = {k: v for k, v in dict(config).items() if not k.startswith("_")}
        c.update(kw)
        return cls(**c)

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, **kw):
        """LOCAL directories only: `<path>[/<subfolder>]/config.json` + `diffusion_pytorch_model.safetensors` (or
        `.bin`, loaded with weights_only=True).  Hub ids are not resolvable here (no network) and raise."""
        d = os.path.join(str(path), subfolder) if subfolder else str(path)
        if not os.path.isdir(d):
            raise OSError(f"{d} is not a local directory: hub model ids cannot be fetched (no network access); "
                          "download the folder and pass its path")
        with open(os.path.join(d, "config.json")) as fh:
            model = cls.from_config(json.load(fh), **kw)
        st = os.path.join(d, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(d, "diffusion_pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd)
        return model.eval()


_UNET_DEFAULTS = dict(
    sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
    mid_block_scale_factor=1, dropout=0.0, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
    transformer_layers_per_block=1, encoder_hid_dim=None, encoder_hid_dim_type=None, attention_head_dim=8,
    num_attention_heads=None, dual_cross_attention=False, use_linear_projection=False, class_embed_type=None,
    addition_embed_type=None, num_class_embeds=None, upcast_attention=False, resnet_time_scale_shift="default",
    time_embedding_type="positional", time_embedding_dim=None, timestep_post_act=None, time_cond_proj_dim=None,
    conv_in_kernel=3, conv_out_kernel=3)


class UNet2DConditionModel(_DiffusersNet):
    """diffusers `UNet2DConditionModel` surface (util.py:1232,1084-1091) over the MI355X lowering."""

    def __init__(self, **kwargs):
        super().__init__()
        unknown = set(kwargs) - set(_UNET_DEFAULTS)
        cfg = dict(_UNET_DEFAULTS)
        cfg.update({k: v for k, v in kwargs.items() if k in _UNET_DEFAULTS})
        if cfg["mid_block_type"] != "UNetMidBlock2DCrossAttn" or cfg["center_input_sample"] or \
                cfg["time_embedding_type"] != "positional" or cfg["conv_out_kernel"] != 3 or \
                cfg["downsample_padding"] != 1 or cfg["mid_block_scale_factor"] != 1:
            raise NotImplementedError("UNet2DConditionModel option outside the SD-2.1 family")
        ubt = list(cfg["up_block_types"])
        want = ["CrossAttnUpBlock2D" if t == "CrossAttnDownBlock2D" else "UpBlock2D"
                for t in reversed(list(cfg["down_block_types"]))]
        if ubt != want:
            raise NotImplementedError(f"up_block_types {ubt} do not mirror down_block_types")
        geo = _geometry(cfg)
        net = nets.DiffusionModelUNet(out_channels=cfg["out_channels"], **geo)
        self._setup(net, cfg)
        self._ignored_kwargs = sorted(unknown)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                down_intrablock_additional_residuals=None, encoder_attention_mask=None, return_dict: bool = True):
        _check_unused(timestep_cond=timestep_cond, attention_mask=attention_mask,
                      cross_attention_kwargs=cross_attention_kwargs, added_cond_kwargs=added_cond_kwargs,
                      down_intrablock_additional_residuals=down_intrablock_additional_residuals,
                      encoder_attention_mask=encoder_attention_mask)
        self._sync()
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1)
        if self.config.get("num_class_embeds") is None:
            class_labels = None  # diffusers ignores class_labels when the model has no class embedding
        out = self._net(x=sample, timesteps=t, context=encoder_hidden_states, class_labels=class_labels,
                        down_block_additional_residuals=down_block_additional_residuals,
                        mid_block_additional_residual=mid_block_additional_residual)
        return UNet2DConditionOutput(sample=out) if return_dict else (out,)


_CN_DEFAULTS = dict(
    in_channels=4, conditioning_channels=3, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn", only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32,
    norm_eps=1e-5, cross_attention_dim=1280, transformer_layers_per_block=1, encoder_hid_dim=None,
    encoder_hid_dim_type=None, attention_head_dim=8, num_attention_heads=None, use_linear_projection=False,
    class_embed_type=None, addition_embed_type=None, addition_time_embed_dim=None, num_class_embeds=None,
    upcast_attention=False, resnet_time_scale_shift="default", projection_class_embeddings_input_dim=None,
    controlnet_conditioning_channel_order="rgb", conditioning_embedding_out_channels=(16, 32, 96, 256),
    global_pool_conditions=False, addition_embed_type_num_heads=64, time_embedding_dim=None)


class ControlNetModel(_DiffusersNet):
    """diffusers `ControlNetModel` surface (util.py:1342-1352, 953-961) over the MI355X lowering."""

    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(_CN_DEFAULTS)
        cfg.update({k: v for k, v in kwargs.items() if k in _CN_DEFAULTS})
        if cfg["mid_block_type"] != "UNetMidBlock2DCrossAttn" or cfg["global_pool_conditions"] or \
                cfg["downsample_padding"] != 1 or cfg["mid_block_scale_factor"] != 1:
            raise NotImplementedError("ControlNetModel option outside the SD-2.1 family")
        if cfg["controlnet_conditioning_channel_order"] not in ("rgb", "bgr"):
            raise ValueError(f"unknown `controlnet_conditioning_channel_order`: "
                             f"{cfg['controlnet_conditioning_channel_order']}")
        geo = _geometry(cfg)
        net = nets.ControlNet(conditioning_embedding_in_channels=cfg["conditioning_channels"],
                              conditioning_embedding_num_channels=tuple(cfg["conditioning_embedding_out_channels"]),
                              **geo)
        self._setup(net, cfg)

    @classmethod
    def from_unet(cls, unet: UNet2DConditionModel, controlnet_conditioning_channel_order: str = "rgb",
                  conditioning_embedding_out_channels: Sequence[int] = (16, 32, 96, 256),
                  load_weights_from_unet: bool = True, conditioning_channels: int = 3):
        """Same geometry as `unet`, encoder half initialised from its weights (util.py:1342)."""
        c = unet.config
        keys = ("in_channels", "flip_sin_to_cos", "freq_shift", "down_block_types", "only_cross_attention",
                "block_out_channels", "layers_per_block", "downsample_padding", "mid_block_scale_factor", "act_fn",
                "norm_num_groups", "norm_eps", "cross_attention_dim", "transformer_layers_per_block", "encoder_hid_dim",
                "encoder_hid_dim_type", "attention_head_dim", "num_attention_heads", "use_linear_projection",
                "class_embed_type", "addition_embed_type", "num_class_embeds", "upcast_attention",
                "resnet_time_scale_shift", "mid_block_type", "time_embedding_dim")
        cn = cls(**{k: c[k] for k in keys if k in c}, conditioning_channels=conditioning_channels,
                 controlnet_conditioning_channel_order=controlnet_conditioning_channel_order,
                 conditioning_embedding_out_channels=tuple(conditioning_embedding_out_channels))
        if load_weights_from_unet:
            src = unet._net.state_dict()
            own = cn._net.state_dict()
            pref = ("conv_in.", "time_embed.", "class_embedding.", "down_blocks.", "middle_block.")
            cn._net.load_state_dict({k: (src[k] if k.startswith(pref) and k in src else v) for k, v in own.items()})
        return cn.to(unet.device)

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale: float = 1.0,
                class_labels=None, timestep_cond=None, attention_mask=None, added_cond_kwargs=None,
                cross_attention_kwargs=None, guess_mode: bool = False, return_dict: bool = True):
        _check_unused(timestep_cond=timestep_cond, attention_mask=attention_mask, added_cond_kwargs=added_cond_kwargs,
                      cross_attention_kwargs=cross_attention_kwargs, guess_mode=guess_mode)
        self._sync()
        if self.config.controlnet_conditioning_channel_order == "bgr":
            controlnet_cond = torch.flip(controlnet_cond, dims=[1])
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1)
        if self.config.get("num_class_embeds") is None:
            class_labels = None
        down, mid = self._net(x=sample, timesteps=t, controlnet_cond=controlnet_cond,
                              conditioning_scale=conditioning_scale, context=encoder_hidden_states,
                              class_labels=class_labels)
        if not return_dict:
            return tuple(down), mid
        return ControlNetOutput(down_block_res_samples=tuple(down), mid_block_res_sample=mid)


def load_ccnet_checkpoint(path: str, model: nn.Module, key: str):
    """util.py:1305-1323 / 1364-1379: `ckpt[key]` (key = "diffusion" | "controlnet" | "state_dict"), falling back to the
    file being the state dict itself.  Loaded with weights_only=True (nothing in the file is executed)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    sd = ckpt[key] if isinstance(ckpt, dict) and key in ckpt and isinstance(ckpt[key], dict) else ckpt
    model.load_state_dict(sd)
    return model


# ------------------------------------------------------------------------------------------------ SD-VAE
@dataclass
class DecoderOutput:
    sample: torch.Tensor


class DiagonalGaussianDistribution:
    """Posterior of `AutoencoderKL.encode` (what `Stage1Wrapper.forward` samples, util.py:1484-1487)."""

    def __init__(self, mean: torch.Tensor, std: torch.Tensor):
        self.mean, self.std = mean, std
        self.var = std * std
        self.logvar = 2.0 * torch.log(std)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * eps

    def mode(self) -> torch.Tensor:
        return self.mean

    def kl(self) -> torch.Tensor:
        return 0.5 * torch.sum(self.mean ** 2 + self.var - 1.0 - self.logvar, dim=[1, 2, 3])


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


_VAE_DEFAULTS = dict(
    in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",), up_block_types=("UpDecoderBlock2D",),
    block_out_channels=(64,), layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=32, sample_size=32,
    scaling_factor=0.18215, shift_factor=None, latents_mean=None, latents_std=None, force_upcast=True,
    use_quant_conv=True, use_post_quant_conv=True, mid_block_add_attention=True)

_RES_D2M = {"conv1.": "conv1.conv.", "conv2.": "conv2.conv.", "conv_shortcut.": "nin_shortcut.conv."}
_ATT_D2M = {"group_norm.": "norm.", "to_out.0.": "proj_attn.",
            # pre-0.18 checkpoints of the same block
            "query.": "to_q.", "key.": "to_k.", "value.": "to_v."}


class AutoencoderKL(_DiffusersNet):
    """diffusers `AutoencoderKL` (the SD-VAE the reference loads with `AutoencoderKL.from_pretrained(source_model,
    subfolder="vae")`, util.py:1215,1227) over the lowered autoencoder: `encode(x).latent_dist`, `decode(z).sample`,
    diffusers-named state dict (`encoder.down_blocks.i.resnets.j...`, `quant_conv`, ...)."""

    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(_VAE_DEFAULTS)
        cfg.update({k: v for k, v in kwargs.items() if k in _VAE_DEFAULTS})
        boc = list(cfg["block_out_channels"])
        n = len(boc)
        if list(cfg["down_block_types"]) != ["DownEncoderBlock2D"] * n or \
                list(cfg["up_block_types"]) != ["UpDecoderBlock2D"] * n or cfg["act_fn"] not in ("silu", "swish") or \
                not cfg["use_quant_conv"] or not cfg["use_post_quant_conv"]:
            raise NotImplementedError("AutoencoderKL option outside the SD-VAE family")
        lpb, lat = cfg["layers_per_block"], cfg["latent_channels"]
        mid = bool(cfg["mid_block_add_attention"])
        net = nets.AutoencoderKL(
            spatial_dims=2, in_channels=cfg["in_channels"], out_channels=cfg["out_channels"], num_res_blocks=lpb,
            num_channels=boc, attention_levels=[False] * n, latent_channels=lat,
            norm_num_groups=cfg["norm_num_groups"], norm_eps=1e-6, with_encoder_nonlocal_attn=mid,
            with_decoder_nonlocal_attn=mid, _decoder_num_res_blocks=lpb + 1, _final_act=True,
            _encoder_out_channels=2 * lat)
        if not mid:
            raise NotImplementedError("mid_block_add_attention=False")
        object.__setattr__(self, "config", FrozenConfig(cfg))
        self._net = net
        # (diffusers prefix, lowered prefix, kind) for every block of the flat encoder / decoder lists
        m: List[Tuple[str, str, str]] = []
        for side, nres, blocks in (("encoder", lpb, net.encoder.blocks), ("decoder", lpb + 1, net.decoder.blocks)):
            i = 0
            m.append((f"{side}.conv_in.", f"{side}.blocks.0.conv.", "conv"))
            i = 1
            if side == "decoder":
                for nm, kind in (("resnets.0", "res"), ("attentions.0", "attn"), ("resnets.1", "res")):
                    m.append((f"decoder.mid_block.{nm}.", f"decoder.blocks.{i}.", kind))
                    i += 1
            grp = "down_blocks" if side == "encoder" else "up_blocks"
            smp = "downsamplers" if side == "encoder" else "upsamplers"
            for lvl in range(n):
                for j in range(nres):
                    m.append((f"{side}.{grp}.{lvl}.resnets.{j}.", f"{side}.blocks.{i}.", "res"))
                    i += 1
                if lvl != n - 1:
                    m.append((f"{side}.{grp}.{lvl}.{smp}.0.conv.", f"{side}.blocks.{i}.conv.conv.", "conv"))
                    i += 1
            if side == "encoder":
                for nm, kind in (("resnets.0", "res"), ("attentions.0", "attn"), ("resnets.1", "res")):
                    m.append((f"encoder.mid_block.{nm}.", f"encoder.blocks.{i}.", kind))
                    i += 1
            m.append((f"{side}.conv_norm_out.", f"{side}.blocks.{i}.", "norm"))
            m.append((f"{side}.conv_out.", f"{side}.blocks.{i + 1}.conv.", "conv"))
            assert i + 2 == len(blocks)
        m.append(("post_quant_conv.", "post_quant_conv.conv.", "conv"))
        self._map = m

    # ---- state dict in diffusers names
    def _leaf(self, rest: str, kind: str, to_monai: bool) -> str:
        table = _RES_D2M if kind == "res" else _ATT_D2M if kind == "attn" else {}
        for d, mm in table.items():
            a, b = (d, mm) if to_monai else (mm, d)
            if rest.startswith(a) and (to_monai or d in ("conv1.", "conv2.", "conv_shortcut.", "group_norm.", "to_out.0.")):
                return b + rest[len(a):]
        return rest

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = self._net.state_dict(keep_vars=keep_vars)
        out = destination if destination is not None else {}
        for dp, mp, kind in self._map:
            for k, v in sd.items():
                if k.startswith(mp):
                    out[prefix + dp + self._leaf(k[len(mp):], kind, False)] = v
        for leaf in ("weight", "bias"):  # quant_conv: rows [mean | logvar]
            out[prefix + "quant_conv." + leaf] = torch.cat([sd[f"quant_conv_mu.conv.{leaf}"],
                                                            sd[f"quant_conv_log_sigma.conv.{leaf}"]], 0)
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd, out = dict(state_dict), {}
        lat = self.config.latent_channels
        for leaf in ("weight", "bias"):
            q = sd.pop(f"quant_conv.{leaf}", None)
            if q is not None:
                out[f"quant_conv_mu.conv.{leaf}"], out[f"quant_conv_log_sigma.conv.{leaf}"] = q[:lat], q[lat:]
        for k, v in sd.items():
            for dp, mp, kind in self._map:
                if k.startswith(dp):
                    mk = mp + self._leaf(k[len(dp):], kind, True)
                    if kind == "attn" and v.dim() == 4:  # very old checkpoints stored the attention Linears as 1x1 convs
                        v = v[:, :, 0, 0]
                    out[mk] = v
                    break
            else:
                out[k] = v  # unknown key: let nn.Module report it
        return self._net.load_state_dict(out, strict=strict, **kw)

    # ---- diffusers call surface
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        mean, std = self._net.encode(x)
        post = DiagonalGaussianDistribution(mean, std)
        return AutoencoderKLOutput(latent_dist=post) if return_dict else (post,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        dec = self._net.decode(z)
        return DecoderOutput(sample=dec) if return_dict else (dec,)

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True,
                generator: Optional[torch.Generator] = None):
        post = self.encode(sample).latent_dist
        z = post.sample(generator=generator) if sample_posterior else post.mode()
        return self.decode(z, return_dict=return_dict)
