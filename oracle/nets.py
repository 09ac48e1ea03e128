"""fp32 PyTorch restatement of the monai-generative networks the reference builds.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).

Reference call sites this follows:
  * DiffusionModelUNet(**config["ldm"]["params"])     -- util.py:1297
  * ControlNet(**config["controlnet"]["params"])      -- util.py:1356
  * AutoencoderKL(**config["stage1"]["params"])       -- util.py:1268
  * adjust_class_embedding (continuous acq-time MLP)  -- util.py:1094-1132
  * forward kwargs                                    -- util.py:944-951, 1054-1069
Upstream algorithm: Project-MONAI/GenerativeModels generative/networks/nets/
{diffusion_model_unet,controlnet,autoencoderkl}.py as recalled in SURVEY.md
Appendix A (A.0-A.5, A.7); state-dict key names follow Appendix A.8.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


def _tuple(v, n):
    if isinstance(v, (list, tuple)):
        return tuple(v)
    return tuple([v] * n)


class Convolution(nn.Sequential):
    """monai.networks.blocks.Convolution(conv_only=True) == Sequential(conv=nn.Conv2d) (A.0)."""

    def __init__(self, cin, cout, kernel_size=3, strides=1, padding=1):
        super().__init__()
        self.add_module("conv", nn.Conv2d(cin, cout, kernel_size, strides, padding, bias=True))


def zero_module(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


def get_timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """A.1: cos first, then sin."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    freqs = torch.exp(exponent / half)
    args = timesteps[:, None].float() * freqs[None, :]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class CrossAttention(nn.Module):
    """A.3: heads folded into batch, baddbmm(scale) -> softmax -> bmm."""

    def __init__(self, query_dim, cross_attention_dim=None, num_attention_heads=8, num_head_channels=64):
        super().__init__()
        inner = num_head_channels * num_attention_heads
        cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.scale = 1.0 / math.sqrt(num_head_channels)
        self.num_heads = num_attention_heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_attention_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_attention_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))

    def _heads_to_batch(self, x):
        b, s, d = x.shape
        h = self.num_heads
        return x.reshape(b, s, h, d // h).permute(0, 2, 1, 3).reshape(b * h, s, d // h)

    def _batch_to_heads(self, x):
        bh, s, d = x.shape
        h = self.num_heads
        return x.reshape(bh // h, h, s, d).permute(0, 2, 1, 3).reshape(bh // h, s, d * h)

    def forward(self, x, context=None):
        q = self.to_q(x)
        context = context if context is not None else x
        k = self.to_k(context)
        v = self.to_v(context)
        q, k, v = self._heads_to_batch(q), self._heads_to_batch(k), self._heads_to_batch(v)
        scores = torch.baddbmm(
            torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
            q, k.transpose(-1, -2), beta=0, alpha=self.scale)
        probs = scores.softmax(dim=-1)
        out = torch.bmm(probs, v)
        return self.to_out(self._batch_to_heads(out))


class MLPBlockGEGLU(nn.Module):
    """monai MLPBlock(hidden, mlp_dim, act="GEGLU"): linear1 C->2*mlp, a*gelu(g), linear2 mlp->C (A.3)."""

    def __init__(self, hidden, mlp_dim):
        super().__init__()
        self.linear1 = nn.Linear(hidden, mlp_dim * 2)
        self.linear2 = nn.Linear(mlp_dim, hidden)

    def forward(self, x):
        a, g = self.linear1(x).chunk(2, dim=-1)
        return self.linear2(a * F.gelu(g))


class BasicTransformerBlock(nn.Module):
    def __init__(self, num_channels, num_attention_heads, num_head_channels, cross_attention_dim=None):
        super().__init__()
        self.attn1 = CrossAttention(num_channels, None, num_attention_heads, num_head_channels)
        self.ff = MLPBlockGEGLU(num_channels, num_channels * 4)
        self.attn2 = CrossAttention(num_channels, cross_attention_dim, num_attention_heads, num_head_channels)
        self.norm1 = nn.LayerNorm(num_channels)
        self.norm2 = nn.LayerNorm(num_channels)
        self.norm3 = nn.LayerNorm(num_channels)

    def forward(self, x, context=None):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context=context) + x
        x = self.ff(self.norm3(x)) + x
        return x


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, num_attention_heads, num_head_channels, num_layers=1,
                 norm_num_groups=32, norm_eps=1e-6, cross_attention_dim=None):
        super().__init__()
        inner = num_attention_heads * num_head_channels
        self.norm = nn.GroupNorm(norm_num_groups, in_channels, eps=norm_eps, affine=True)
        self.proj_in = Convolution(in_channels, inner, kernel_size=1, padding=0)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner, num_attention_heads, num_head_channels, cross_attention_dim)
            for _ in range(num_layers)])
        self.proj_out = zero_module(Convolution(inner, in_channels, kernel_size=1, padding=0))

    def forward(self, x, context=None):
        b, c, h, w = x.shape
        residual = x
        x = self.norm(x)
        x = self.proj_in(x)
        inner = x.shape[1]
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        for blk in self.transformer_blocks:
            x = blk(x, context=context)
        x = x.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        x = self.proj_out(x)
        return x + residual


class ResnetBlock(nn.Module):
    """A.2."""

    def __init__(self, in_channels, temb_channels, out_channels=None, norm_num_groups=32, norm_eps=1e-6):
        super().__init__()
        out_channels = out_channels or in_channels
        self.norm1 = nn.GroupNorm(norm_num_groups, in_channels, eps=norm_eps, affine=True)
        self.conv1 = Convolution(in_channels, out_channels)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(norm_num_groups, out_channels, eps=norm_eps, affine=True)
        self.conv2 = zero_module(Convolution(out_channels, out_channels))
        if in_channels == out_channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = Convolution(in_channels, out_channels, kernel_size=1, padding=0)

    def forward(self, x, emb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(emb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return self.skip_connection(x) + h


class Downsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.op = Convolution(channels, channels, kernel_size=3, strides=2, padding=1)

    def forward(self, x, emb=None):
        return self.op(x)


class Upsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = Convolution(channels, channels)

    def forward(self, x, emb=None):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, n_res, add_down, with_attn, heads_ch, cross_dim, groups, eps, layers):
        super().__init__()
        self.resnets = nn.ModuleList()
        atts = []
        for i in range(n_res):
            self.resnets.append(ResnetBlock(cin if i == 0 else cout, temb, cout, groups, eps))
            if with_attn:
                atts.append(SpatialTransformer(cout, cout // heads_ch, heads_ch, layers, groups, eps, cross_dim))
        if with_attn:
            self.attentions = nn.ModuleList(atts)
        self.with_attn = with_attn
        self.downsampler = Downsample(cout) if add_down else None

    def forward(self, h, emb, context):
        outs = []
        for i, r in enumerate(self.resnets):
            h = r(h, emb)
            if self.with_attn:
                h = self.attentions[i](h, context)
            outs.append(h)
        if self.downsampler is not None:
            h = self.downsampler(h)
            outs.append(h)
        return h, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb, heads_ch, cross_dim, groups, eps, layers):
        super().__init__()
        self.resnet_1 = ResnetBlock(ch, temb, ch, groups, eps)
        self.attention = SpatialTransformer(ch, ch // heads_ch, heads_ch, layers, groups, eps, cross_dim)
        self.resnet_2 = ResnetBlock(ch, temb, ch, groups, eps)

    def forward(self, h, emb, context):
        h = self.resnet_1(h, emb)
        h = self.attention(h, context)
        return self.resnet_2(h, emb)


class UpBlock(nn.Module):
    def __init__(self, cin, prev, cout, temb, n_res, add_up, with_attn, heads_ch, cross_dim, groups, eps, layers):
        super().__init__()
        self.resnets = nn.ModuleList()
        atts = []
        for i in range(n_res):
            skip = cin if i == n_res - 1 else cout
            rin = prev if i == 0 else cout
            self.resnets.append(ResnetBlock(rin + skip, temb, cout, groups, eps))
            if with_attn:
                atts.append(SpatialTransformer(cout, cout // heads_ch, heads_ch, layers, groups, eps, cross_dim))
        if with_attn:
            self.attentions = nn.ModuleList(atts)
        self.with_attn = with_attn
        self.upsampler = Upsample(cout) if add_up else None

    def forward(self, h, skips, emb, context):
        for i, r in enumerate(self.resnets):
            s = skips[-1]
            skips = skips[:-1]
            h = r(torch.cat([h, s], dim=1), emb)
            if self.with_attn:
                h = self.attentions[i](h, context)
        if self.upsampler is not None:
            h = self.upsampler(h)
        return h


class _UNetTrunk(nn.Module):
    """conv_in, time/class embedding, down_blocks and middle_block shared by A.4 and A.5."""

    def _build_trunk(self, in_channels, num_res_blocks, num_channels, attention_levels, norm_num_groups,
                     norm_eps, num_head_channels, with_conditioning, transformer_num_layers,
                     cross_attention_dim, num_class_embeds):
        n = len(num_channels)
        self.block_out_channels = list(num_channels)
        self.num_res_blocks_ = _tuple(num_res_blocks, n)
        self.num_head_channels_ = _tuple(num_head_channels, n)
        self.attention_levels_ = tuple(attention_levels)
        if with_conditioning and cross_attention_dim is None:
            raise ValueError("with_conditioning=True needs cross_attention_dim")
        if any(attention_levels) and not with_conditioning:
            raise NotImplementedError("oracle restates the with_conditioning=True path the reference uses")
        self.conv_in = Convolution(in_channels, num_channels[0])
        ted = num_channels[0] * 4
        self.time_embed = nn.Sequential(nn.Linear(num_channels[0], ted), nn.SiLU(), nn.Linear(ted, ted))
        self.num_class_embeds = num_class_embeds
        if num_class_embeds is not None:
            self.class_embedding = nn.Embedding(num_class_embeds, ted)
        self.down_blocks = nn.ModuleList()
        out_ch = num_channels[0]
        for i in range(n):
            in_ch, out_ch = out_ch, num_channels[i]
            self.down_blocks.append(DownBlock(
                in_ch, out_ch, ted, self.num_res_blocks_[i], i != n - 1,
                bool(attention_levels[i]) and with_conditioning, self.num_head_channels_[i],
                cross_attention_dim, norm_num_groups, norm_eps, transformer_num_layers))
        self.middle_block = MidBlock(num_channels[-1], ted, self.num_head_channels_[-1], cross_attention_dim,
                                     norm_num_groups, norm_eps, transformer_num_layers)
        return ted

    def _embed(self, x, timesteps, class_labels):
        t_emb = get_timestep_embedding(timesteps, self.block_out_channels[0]).to(dtype=x.dtype)
        emb = self.time_embed(t_emb)
        if self.num_class_embeds is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            emb = emb + self.class_embedding(class_labels).to(dtype=x.dtype)
        return emb


class DiffusionModelUNet(_UNetTrunk):
    """A.4."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int,
                 num_res_blocks=(2, 2, 2, 2), num_channels=(32, 64, 64, 64),
                 attention_levels=(False, False, True, True), norm_num_groups: int = 32,
                 norm_eps: float = 1e-6, resblock_updown: bool = False, num_head_channels=8,
                 with_conditioning: bool = False, transformer_num_layers: int = 1,
                 cross_attention_dim: Optional[int] = None, num_class_embeds: Optional[int] = None,
                 upcast_attention: bool = False, use_flash_attention: bool = False):
        super().__init__()
        assert spatial_dims == 2 and not resblock_updown
        ted = self._build_trunk(in_channels, num_res_blocks, num_channels, attention_levels, norm_num_groups,
                                norm_eps, num_head_channels, with_conditioning, transformer_num_layers,
                                cross_attention_dim, num_class_embeds)
        n = len(num_channels)
        rev_ch = list(reversed(num_channels))
        rev_res = list(reversed(self.num_res_blocks_))
        rev_att = list(reversed(self.attention_levels_))
        rev_heads = list(reversed(self.num_head_channels_))
        self.up_blocks = nn.ModuleList()
        out_ch = rev_ch[0]
        for i in range(n):
            prev, out_ch = out_ch, rev_ch[i]
            in_ch = rev_ch[min(i + 1, n - 1)]
            self.up_blocks.append(UpBlock(
                in_ch, prev, out_ch, ted, rev_res[i] + 1, i != n - 1,
                bool(rev_att[i]) and with_conditioning, rev_heads[i], cross_attention_dim,
                norm_num_groups, norm_eps, transformer_num_layers))
        self.out = nn.Sequential(
            nn.GroupNorm(norm_num_groups, num_channels[0], eps=norm_eps, affine=True), nn.SiLU(),
            zero_module(Convolution(num_channels[0], out_channels)))

    def forward(self, x, timesteps, context=None, class_labels=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None):
        emb = self._embed(x, timesteps, class_labels)
        h = self.conv_in(x)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, emb, context)
            skips.extend(outs)
        if down_block_additional_residuals is not None:
            skips = [s + a for s, a in zip(skips, down_block_additional_residuals)]
        h = self.middle_block(h, emb, context)
        if mid_block_additional_residual is not None:
            h = h + mid_block_additional_residual
        for blk in self.up_blocks:
            k = len(blk.resnets)
            res, skips = skips[-k:], skips[:-k]
            h = blk(h, res, emb, context)
        return self.out(h)


class ControlNetConditioningEmbedding(nn.Module):
    def __init__(self, in_channels, out_channels, num_channels=(16, 32, 96, 256)):
        super().__init__()
        self.conv_in = Convolution(in_channels, num_channels[0])
        self.blocks = nn.ModuleList()
        for i in range(len(num_channels) - 1):
            self.blocks.append(Convolution(num_channels[i], num_channels[i]))
            self.blocks.append(Convolution(num_channels[i], num_channels[i + 1], strides=2))
        self.conv_out = zero_module(Convolution(num_channels[-1], out_channels))

    def forward(self, c):
        e = F.silu(self.conv_in(c))
        for b in self.blocks:
            e = F.silu(b(e))
        return self.conv_out(e)


class ControlNet(_UNetTrunk):
    """A.5."""

    def __init__(self, spatial_dims: int, in_channels: int, num_res_blocks=(2, 2, 2, 2),
                 num_channels=(32, 64, 64, 64), attention_levels=(False, False, True, True),
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, resblock_updown: bool = False,
                 num_head_channels=8, with_conditioning: bool = False, transformer_num_layers: int = 1,
                 cross_attention_dim: Optional[int] = None, num_class_embeds: Optional[int] = None,
                 upcast_attention: bool = False, use_flash_attention: bool = False,
                 conditioning_embedding_in_channels: int = 1,
                 conditioning_embedding_num_channels: Sequence[int] = (16, 32, 96, 256)):
        super().__init__()
        assert spatial_dims == 2 and not resblock_updown
        self._build_trunk(in_channels, num_res_blocks, num_channels, attention_levels, norm_num_groups,
                          norm_eps, num_head_channels, with_conditioning, transformer_num_layers,
                          cross_attention_dim, num_class_embeds)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(
            conditioning_embedding_in_channels, num_channels[0], tuple(conditioning_embedding_num_channels))
        n = len(num_channels)
        self.controlnet_down_blocks = nn.ModuleList()
        out_ch = num_channels[0]
        self.controlnet_down_blocks.append(zero_module(nn.Conv2d(out_ch, out_ch, 1)))
        for i in range(n):
            out_ch = num_channels[i]
            for _ in range(self.num_res_blocks_[i]):
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(out_ch, out_ch, 1)))
            if i != n - 1:
                self.controlnet_down_blocks.append(zero_module(nn.Conv2d(out_ch, out_ch, 1)))
        self.controlnet_mid_block = zero_module(nn.Conv2d(num_channels[-1], num_channels[-1], 1))

    def forward(self, x, timesteps, controlnet_cond, conditioning_scale: float = 1.0, context=None,
                class_labels=None):
        emb = self._embed(x, timesteps, class_labels)
        h = self.conv_in(x)
        h = h + self.controlnet_cond_embedding(controlnet_cond)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, emb, context)
            skips.extend(outs)
        h = self.middle_block(h, emb, context)
        down = [z(s) * conditioning_scale for s, z in zip(skips, self.controlnet_down_blocks)]
        mid = self.controlnet_mid_block(h) * conditioning_scale
        return down, mid


# --------------------------------------------------------------------------- AutoencoderKL (A.7)

class AEResBlock(nn.Module):
    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps, affine=True)
        self.conv1 = Convolution(cin, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps, affine=True)
        self.conv2 = Convolution(cout, cout)
        self.nin_shortcut = Convolution(cin, cout, kernel_size=1, padding=0) if cin != cout else nn.Identity()

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return self.nin_shortcut(x) + h


class AEAttentionBlock(nn.Module):
    """Single-head (num_head_channels=None) spatial self-attention with residual."""

    def __init__(self, ch, groups, eps):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=eps, affine=True)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.proj_attn = nn.Linear(ch, ch)
        self.scale = 1.0 / math.sqrt(ch)

    def forward(self, x):
        b, c, h, w = x.shape
        r = x
        t = self.norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * self.scale, dim=-1)
        o = self.proj_attn(torch.bmm(p, v))
        return o.transpose(1, 2).reshape(b, c, h, w) + r


class AEDownsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = Convolution(ch, ch, kernel_size=3, strides=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0.0))


class AEUpsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = Convolution(ch, ch)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _SiLUAfter(nn.Module):
    """GroupNorm followed by SiLU (diffusers Encoder/Decoder: conv_norm_out -> conv_act); keeps the GroupNorm's
    parameter names (`weight`, `bias`) so the state dict does not change."""

    def __init__(self, groups, ch, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))
        self.groups, self.eps = groups, eps

    def forward(self, x):
        return F.silu(F.group_norm(x, self.groups, self.weight, self.bias, self.eps))


class AEEncoder(nn.Module):
    def __init__(self, in_channels, num_channels, out_channels, num_res_blocks, groups, eps, attention_levels,
                 with_nonlocal_attn, final_act=False):
        super().__init__()
        blocks = [Convolution(in_channels, num_channels[0])]
        out_ch = num_channels[0]
        for i in range(len(num_channels)):
            in_ch, out_ch = out_ch, num_channels[i]
            for _ in range(num_res_blocks[i]):
                blocks.append(AEResBlock(in_ch, out_ch, groups, eps))
                in_ch = out_ch
                if attention_levels[i]:
                    blocks.append(AEAttentionBlock(in_ch, groups, eps))
            if i != len(num_channels) - 1:
                blocks.append(AEDownsample(in_ch))
        if with_nonlocal_attn:
            blocks += [AEResBlock(num_channels[-1], num_channels[-1], groups, eps),
                       AEAttentionBlock(num_channels[-1], groups, eps),
                       AEResBlock(num_channels[-1], num_channels[-1], groups, eps)]
        blocks.append(_SiLUAfter(groups, num_channels[-1], eps) if final_act else
                      nn.GroupNorm(groups, num_channels[-1], eps=eps, affine=True))
        blocks.append(Convolution(num_channels[-1], out_channels))
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class AEDecoder(nn.Module):
    def __init__(self, num_channels, in_channels, out_channels, num_res_blocks, groups, eps, attention_levels,
                 with_nonlocal_attn, final_act=False):
        super().__init__()
        rev = list(reversed(num_channels))
        blocks = [Convolution(in_channels, rev[0])]
        if with_nonlocal_attn:
            blocks += [AEResBlock(rev[0], rev[0], groups, eps), AEAttentionBlock(rev[0], groups, eps),
                       AEResBlock(rev[0], rev[0], groups, eps)]
        rev_att = list(reversed(attention_levels))
        rev_res = list(reversed(num_res_blocks))
        out_ch = rev[0]
        for i in range(len(rev)):
            in_ch, out_ch = out_ch, rev[i]
            for _ in range(rev_res[i]):
                blocks.append(AEResBlock(in_ch, out_ch, groups, eps))
                in_ch = out_ch
                if rev_att[i]:
                    blocks.append(AEAttentionBlock(in_ch, groups, eps))
            if i != len(rev) - 1:
                blocks.append(AEUpsample(in_ch))
        blocks.append(_SiLUAfter(groups, in_ch, eps) if final_act else nn.GroupNorm(groups, in_ch, eps=eps, affine=True))
        blocks.append(Convolution(in_ch, out_channels))
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class AutoencoderKL(nn.Module):
    def __init__(self, spatial_dims: int, in_channels: int = 1, out_channels: int = 1, num_res_blocks=(2, 2, 2, 2),
                 num_channels=(32, 64, 64, 64), attention_levels=(False, False, True, True),
                 latent_channels: int = 3, norm_num_groups: int = 32, norm_eps: float = 1e-6,
                 with_encoder_nonlocal_attn: bool = True, with_decoder_nonlocal_attn: bool = True,
                 use_flash_attention: bool = False, _decoder_num_res_blocks=None, _final_act: bool = False,
                 _encoder_out_channels=None):
        """Underscore options: the diffusers SD-VAE (`diffusers.models.autoencoders.vae.Encoder/Decoder`, 0.32.0) on the
        same block list -- decoder levels with layers_per_block + 1 resnets, SiLU after conv_norm_out, an encoder
        emitting 2 x latent_channels moments that `quant_conv` (here its mean rows / logvar rows) maps to the posterior."""
        super().__init__()
        assert spatial_dims == 2
        n = len(num_channels)
        nrb = _tuple(num_res_blocks, n)
        dnrb = _tuple(_decoder_num_res_blocks, n) if _decoder_num_res_blocks is not None else nrb
        att = _tuple(attention_levels, n)
        enc_out = _encoder_out_channels or latent_channels
        self.num_channels = list(num_channels)
        self.latent_channels = latent_channels
        self.encoder = AEEncoder(in_channels, num_channels, enc_out, nrb, norm_num_groups, norm_eps,
                                 att, with_encoder_nonlocal_attn, _final_act)
        self.decoder = AEDecoder(num_channels, latent_channels, out_channels, dnrb, norm_num_groups, norm_eps,
                                 att, with_decoder_nonlocal_attn, _final_act)
        self.quant_conv_mu = Convolution(enc_out, latent_channels, kernel_size=1, padding=0)
        self.quant_conv_log_sigma = Convolution(enc_out, latent_channels, kernel_size=1, padding=0)
        self.post_quant_conv = Convolution(latent_channels, latent_channels, kernel_size=1, padding=0)

    def encode(self, x):
        h = self.encoder(x)
        z_mu = self.quant_conv_mu(h)
        z_log_var = torch.clamp(self.quant_conv_log_sigma(h), -30.0, 20.0)
        return z_mu, torch.exp(z_log_var / 2)

    def sampling(self, z_mu, z_sigma, eps=None):
        eps = torch.randn_like(z_sigma) if eps is None else eps
        return z_mu + eps * z_sigma

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    def forward(self, x):
        z_mu, z_sigma = self.encode(x)
        z = self.sampling(z_mu, z_sigma)
        return self.decode(z), z_mu, z_sigma


def adjust_class_embedding(model: nn.Module) -> None:
    """util.py:1094-1132 -- replace the class nn.Embedding by Linear(1,ted)-SiLU-Linear(ted,ted)."""
    if not hasattr(model, "block_out_channels"):
        raise ValueError(f"{model.__class__.__name__} has no block_out_channels")
    model.num_class_embeds = int(99999)
    ted = model.block_out_channels[0] * 4
    model.class_embedding = nn.Sequential(nn.Linear(1, ted), nn.SiLU(), nn.Linear(ted, ted))
