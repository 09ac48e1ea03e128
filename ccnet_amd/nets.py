"""MI355X-native DiffusionModelUNet / ControlNet / AutoencoderKL with the reference's L2 call surface.

Drop-in boundary:
  * `DiffusionModelUNet(**config["ldm"]["params"])`, `.forward(x, timesteps, context, class_labels=None,
    down_block_additional_residuals=None, mid_block_additional_residual=None)`      -- util.py:1297, 1054-1069
  * `ControlNet(**config["controlnet"]["params"])`, `.forward(x, timesteps, controlnet_cond,
    conditioning_scale=1.0, context=None, class_labels=None) -> (list, mid)`        -- util.py:1356, 944-951
  * `AutoencoderKL(**config["stage1"]["params"])`, `.encode/.sampling/.decode/.forward` -- util.py:1268, 1490-1491
  * `adjust_class_embedding(model)`                                                 -- util.py:1094-1132
  * `.state_dict()/.load_state_dict()` with monai-generative key names; class names
    contain "DiffusionModelUNet" / "ControlNet" (not "ControlNetModel") because the reference dispatches on
    them (util.py:943, 1051, 1100).

Architecture: the nn.Module tree below only HOLDS parameters (standard torch layers are used as containers so
default initialisation and key names match upstream); nothing in it has a torch forward.  `forward` lowers the
network to a sequence of hand-written gfx950 kernels (ccnet_amd.ops -> libccnet_hip.so): activations are NHWC
bf16 resident in HBM, weights are packed once to bf16 [Cout][kr][ks][Cin], GroupNorm/softmax/embedding
statistics stay in fp32.  There is no CPU or eager fallback: without the HIP library every forward raises.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops
from .lib import CcnetError, CcnetUnsupported
from .ops import Act

BF16 = torch.bfloat16
# CCNET_LN_FOLD=1: the transformer blocks' LayerNorms are folded into the GEMMs that consume them (ops.fold_layernorm +
# ops.row_stats: the normalised rows never reach HBM, three launches per block become statistics-only passes).  Measured on
# the whole pipeline (tools/ab_lnfold.sh, same box, batch 16): 2227 / 2231 ms per batch with the fold against 2220 / 2221
# without -- the write it saves is served from the Infinity Cache anyway and the GEMM's accumulator start gets heavier -- so
# the separate LayerNorm kernel stays the default and the fold is an option (parity-tested either way).
LN_FOLD = os.environ.get("CCNET_LN_FOLD", "0") == "1"
# CCNET_LN_INLINE=1 (default): norm1 -> qkv and norm2 -> attn2.to_q of the 320-channel level run as ONE launch each, the
# streaming pointwise kernel taking the row statistics from the rows it stages anyway (ops.ln_inline_ok); the normalised
# rows are never written.  Other levels / small batches keep the LayerNorm kernel.
LN_INLINE = os.environ.get("CCNET_LN_INLINE", "1") == "1"
# Head dim 64: softmax scale * log2(e) folded into the packed to_q weights (one bf16 rounding of the scaled fp32 weight), the
# attention kernel then runs its prescaled instance (no multiply in front of the exponential).  CCNET_ATTN_PRESCALE=0: A/B.
ATTN_PRESCALE = os.environ.get("CCNET_ATTN_PRESCALE", "1") == "1"
# CCNET_FF_FUSED=1 (default): norm3 -> ff.linear1 -> GEGLU -> ff.linear2 -> + x of the 320-channel level run as ONE launch
# (ccnet_ff_geglu_fused_bf16: the hidden tensor never reaches HBM) where the rows fill the chip.  CCNET_FF_FUSED=0: A/B.
FF_FUSED = os.environ.get("CCNET_FF_FUSED", "1") == "1"
# CCNET_CFG_SHARE=1 (default): under classifier-free guidance the unconditional and the text half of the [uncond x B, text x B]
# batch (run_controlnet_inference.py:219-221: the same latent, conditioning image and acquisition time twice) are the SAME rows
# until the first cross-attention reads the prompt, so conv_in, the first resnet and the first transformer block up to attn2's
# query are evaluated once, on B rows, and duplicated there (a sampler asks for it with `_cfg_shared`).  0: A/B, both halves
# evaluated from the start.
CFG_SHARE = os.environ.get("CCNET_CFG_SHARE", "1") == "1"
# CCNET_XATTN_FUSED=1 (default): norm2 -> attn2.to_q -> cross-attention -> attn2.to_out -> + x of the 320-channel level run as ONE
# launch (ccnet_xattn_fused_bf16: neither the query nor the attention output reaches HBM) where the rows fill the chip; the
# prompt's K / V are laid out for it once per prompt batch (context_kv).  CCNET_XATTN_FUSED=0: A/B.
XATTN_FUSED = os.environ.get("CCNET_XATTN_FUSED", "1") == "1"
FF_FUSED_MIN_ROWS = 32768  # 128-row tiles, one workgroup per CU: below 256 tiles the three launches are faster


def _tuple(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else tuple([v] * n)


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


def _even_chunk(n: int, limit: int) -> int:
    """Images per chunk when `n` images must be processed at most `limit` at a time: the fewest chunks, all (nearly) the same
    size.  Sixteen images under a limit of 15 run as 8 + 8, not 15 + 1: a lone last image would be planned differently by
    the conv dispatch (other tiles, split-K) and so round differently -- results must not depend on where in a batch an
    image sits."""
    if n <= limit:
        return max(n, 1)
    k = (n + limit - 1) // limit
    return (n + k - 1) // k


# --------------------------------------------------------------------------- parameter containers
class _Holder(nn.Module):
    """Parameter container; deliberately has no forward."""

    def forward(self, *a, **k):  # pragma: no cover
        raise CcnetError("parameter container: compute goes through the owning network's HIP lowering")


def _zero(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


def _convolution(cin, cout, k=3, stride=1, pad=1) -> _Holder:
    h = _Holder()
    h.conv = nn.Conv2d(cin, cout, k, stride, pad, bias=True)
    return h


def _cross_attention(query_dim, cross_dim, heads, head_ch) -> _Holder:
    h = _Holder()
    inner = heads * head_ch
    cross_dim = cross_dim if cross_dim is not None else query_dim
    h.to_q = nn.Linear(query_dim, inner, bias=False)
    h.to_k = nn.Linear(cross_dim, inner, bias=False)
    h.to_v = nn.Linear(cross_dim, inner, bias=False)
    h.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))
    return h


def _transformer_block(ch, heads, head_ch, cross_dim) -> _Holder:
    h = _Holder()
    h.attn1 = _cross_attention(ch, None, heads, head_ch)
    ff = _Holder()
    ff.linear1 = nn.Linear(ch, ch * 8)
    ff.linear2 = nn.Linear(ch * 4, ch)
    h.ff = ff
    h.attn2 = _cross_attention(ch, cross_dim, heads, head_ch)
    h.norm1 = nn.LayerNorm(ch)
    h.norm2 = nn.LayerNorm(ch)
    h.norm3 = nn.LayerNorm(ch)
    return h


def _spatial_transformer(ch, heads, head_ch, layers, groups, eps, cross_dim) -> _Holder:
    h = _Holder()
    inner = heads * head_ch
    h.norm = nn.GroupNorm(groups, ch, eps=eps, affine=True)
    h.proj_in = _convolution(ch, inner, 1, 1, 0)
    h.transformer_blocks = nn.ModuleList([_transformer_block(inner, heads, head_ch, cross_dim) for _ in range(layers)])
    h.proj_out = _zero(_convolution(inner, ch, 1, 1, 0))
    h.heads, h.head_ch = heads, head_ch
    return h


def _resnet(cin, temb, cout, groups, eps) -> _Holder:
    h = _Holder()
    h.norm1 = nn.GroupNorm(groups, cin, eps=eps, affine=True)
    h.conv1 = _convolution(cin, cout)
    h.time_emb_proj = nn.Linear(temb, cout)
    h.norm2 = nn.GroupNorm(groups, cout, eps=eps, affine=True)
    h.conv2 = _zero(_convolution(cout, cout))
    h.skip_connection = nn.Identity() if cin == cout else _convolution(cin, cout, 1, 1, 0)
    h.cin, h.cout = cin, cout
    return h


class _Trunk(nn.Module):
    """conv_in + embeddings + down blocks + middle block (shared by UNet and ControlNet)."""

    def _build_trunk(self, in_channels, num_res_blocks, num_channels, attention_levels, norm_num_groups, norm_eps,
                     num_head_channels, with_conditioning, transformer_num_layers, cross_attention_dim,
                     num_class_embeds):
        n = len(num_channels)
        if with_conditioning and cross_attention_dim is None:
            raise ValueError("DiffusionModelUNet expects dimension of the cross-attention conditioning "
                             "(cross_attention_dim) when using with_conditioning.")
        if any(attention_levels) and not with_conditioning:
            raise NotImplementedError("attention without conditioning is not on the ccnet hot path")
        if any(c % norm_num_groups for c in num_channels):
            raise ValueError("DiffusionModelUNet expects all num_channels being multiple of norm_num_groups")
        if any(c % 64 for c in num_channels):
            raise CcnetError("the MI355X lowering needs every entry of num_channels to be a multiple of 64")
        if transformer_num_layers != 1:
            raise NotImplementedError("transformer_num_layers != 1")
        self.in_channels = in_channels
        self.block_out_channels = list(num_channels)
        self.num_res_blocks_ = _tuple(num_res_blocks, n)
        self.num_head_channels_ = _tuple(num_head_channels, n)
        self.attention_levels_ = tuple(bool(a) for a in attention_levels)
        self.norm_num_groups, self.norm_eps = norm_num_groups, norm_eps
        # monai-generative hands the network's norm_eps to SpatialTransformer.norm; diffusers' Transformer2DModel fixes
        # its GroupNorm at 1e-6 (ccnet_amd.diffusers_dialect overrides this attribute)
        self.transformer_norm_eps = norm_eps
        self.with_conditioning = with_conditioning
        self.cross_attention_dim = cross_attention_dim
        c0 = num_channels[0]
        ted = c0 * 4
        self.conv_in = _convolution(in_channels, c0)
        self.time_embed = nn.Sequential(nn.Linear(c0, ted), nn.SiLU(), nn.Linear(ted, ted))
        self.num_class_embeds = num_class_embeds
        if num_class_embeds is not None:
            self.class_embedding = nn.Embedding(num_class_embeds, ted)
        self.down_blocks = nn.ModuleList()
        out_ch = c0
        for i in range(n):
            in_ch, out_ch = out_ch, num_channels[i]
            blk = _Holder()
            blk.resnets = nn.ModuleList()
            att = bool(attention_levels[i]) and with_conditioning
            atts = []
            for j in range(self.num_res_blocks_[i]):
                blk.resnets.append(_resnet(in_ch if j == 0 else out_ch, ted, out_ch, norm_num_groups, norm_eps))
                if att:
                    hc = self.num_head_channels_[i]
                    atts.append(_spatial_transformer(out_ch, out_ch // hc, hc, 1, norm_num_groups, norm_eps,
                                                     cross_attention_dim))
            if att:
                blk.attentions = nn.ModuleList(atts)
            if i != n - 1:
                ds = _Holder()
                ds.op = _convolution(out_ch, out_ch, 3, 2, 1)
                blk.downsampler = ds
            else:
                blk.downsampler = None
            self.down_blocks.append(blk)
        mid = _Holder()
        cl, hc = num_channels[-1], self.num_head_channels_[-1]
        mid.resnet_1 = _resnet(cl, ted, cl, norm_num_groups, norm_eps)
        mid.attention = _spatial_transformer(cl, cl // hc, hc, 1, norm_num_groups, norm_eps, cross_attention_dim)
        mid.resnet_2 = _resnet(cl, ted, cl, norm_num_groups, norm_eps)
        self.middle_block = mid
        self._pk = None
        return ted

    # ---- packed-weight cache: rebuilt whenever parameters may have changed
    def _invalidate(self):
        self._pk = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = dict(state_dict)
        # accept `controlnet_down_blocks.k.conv.weight` as well as the bare-conv spelling (Appendix A.8)
        for k in list(sd.keys()):
            if k.startswith(("controlnet_down_blocks.", "controlnet_mid_block.")) and ".conv." in k:
                sd[k.replace(".conv.", ".")] = sd.pop(k)
        r = super().load_state_dict(sd, strict=strict, **kw)
        self._invalidate()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._pk = None
        return r

    def _resnets_in_order(self) -> List[_Holder]:
        rs = []
        for blk in self.down_blocks:
            rs.extend(blk.resnets)
        rs += [self.middle_block.resnet_1, self.middle_block.resnet_2]
        if hasattr(self, "up_blocks"):
            for blk in self.up_blocks:
                rs.extend(blk.resnets)
        return rs

    def _pack(self):
        dev = self.conv_in.conv.weight.device
        if dev.type != "cuda":
            raise CcnetError("ccnet_amd networks run on the GPU only: call .to('cuda') first (no CPU fallback)")
        pk = {}
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()

        def pconv(conv: nn.Conv2d, cin_pad=None, cout_pad=None):
            w = conv.weight.detach().float()
            co, ci, r, _ = w.shape
            cip, cop = cin_pad or ci, cout_pad or co
            wp = torch.zeros(cop, r, r, cip, dtype=torch.float32, device=w.device)
            wp[:co, :, :, :ci] = w.permute(0, 2, 3, 1)
            b = torch.zeros(cop, dtype=torch.float32, device=w.device)
            if conv.bias is not None:
                b[:co] = conv.bias.detach().float()
            return wp.reshape(cop, r * r * cip).to(BF16).contiguous(), b.contiguous()

        def plin(w: torch.Tensor):
            return w.detach().to(BF16).contiguous()

        def pres(r: _Holder):
            d = {"n1": (f32(r.norm1.weight), f32(r.norm1.bias)), "n2": (f32(r.norm2.weight), f32(r.norm2.bias)),
                 "c1": pconv(r.conv1.conv), "c2": pconv(r.conv2.conv), "cout": r.cout,
                 "skip": None if isinstance(r.skip_connection, nn.Identity) else pconv(r.skip_connection.conv)}
            return d

        def pattn(a: _Holder):
            b = a.transformer_blocks[0]
            inner = a.heads * a.head_ch
            qscale = math.log2(math.e) / math.sqrt(a.head_ch)
            # LayerNorm folded into the linear layer that consumes it (ops.fold_layernorm; the GEMM kernel applies the row
            # statistics): norm1 -> to_q|to_k|to_v, norm2 -> attn2.to_q, norm3 -> ff.linear1 (then the GEGLU interleave)
            f_qkv = f_q2 = f_ff1 = None
            if LN_FOLD:
                dv = lambda t: tuple(x.to(dev) for x in t)
                qkv_w = torch.cat([b.attn1.to_q.weight, b.attn1.to_k.weight, b.attn1.to_v.weight], 0)
                f_qkv = dv(ops.fold_layernorm(qkv_w, None, b.norm1.weight, b.norm1.bias))
                f_q2 = dv(ops.fold_layernorm(b.attn2.to_q.weight, None, b.norm2.weight, b.norm2.bias))
                w1, b1, c1 = ops.fold_layernorm(b.ff.linear1.weight, b.ff.linear1.bias, b.norm3.weight, b.norm3.bias)
                f_ff1 = dv((ops.geglu_interleave(w1), ops.geglu_interleave(b1), ops.geglu_interleave(c1)))
            i_qkv = i_q2 = None
            if LN_INLINE and not LN_FOLD and inner == 320:
                dv = lambda t: tuple(x.to(dev) for x in t)
                qs = qscale if ATTN_PRESCALE and a.head_ch == 64 else 1.0
                i_qkv = dv(ops.fold_layernorm(torch.cat([b.attn1.to_q.weight.detach().float() * qs, b.attn1.to_k.weight.detach().float(),
                                                         b.attn1.to_v.weight.detach().float()], 0), None, b.norm1.weight, b.norm1.bias))
                i_q2 = dv(ops.fold_layernorm(b.attn2.to_q.weight.detach().float() * qs, None, b.norm2.weight, b.norm2.bias))
            ffp = None
            if FF_FUSED and not LN_FOLD:
                ffp = ops.ff_pack(b.ff.linear1.weight.detach().to(dev), b.ff.linear1.bias.detach().to(dev),
                                  b.ff.linear2.weight.detach().to(dev))  # None: width not covered by the fused kernel
            xap = None
            if XATTN_FUSED and not LN_FOLD and a.head_ch == 64 and b.attn2.to_q.bias is None:
                xap = ops.xattn_pack(b.attn2.to_q.weight.detach().to(dev), None, b.attn2.to_out[0].weight.detach().to(dev),
                                     a.heads)  # None: width not covered by the fused kernel
            return {
                "ffp": ffp, "xap": xap,
                "qkv1i": i_qkv, "q2i": i_q2,
                "qkv1f": f_qkv, "q2f": f_q2, "ff1f": f_ff1,
                "norm": (f32(a.norm.weight), f32(a.norm.bias)), "pin": pconv(a.proj_in.conv),
                "pout": pconv(a.proj_out.conv), "heads": a.heads, "head_ch": a.head_ch, "inner": inner,
                "ln1": (f32(b.norm1.weight), f32(b.norm1.bias)), "ln2": (f32(b.norm2.weight), f32(b.norm2.bias)),
                "ln3": (f32(b.norm3.weight), f32(b.norm3.bias)),
                "qkv1": plin(torch.cat([b.attn1.to_q.weight, b.attn1.to_k.weight, b.attn1.to_v.weight], 0)),
                # head dim 64: the softmax scale (and the log2(e) of the kernel's exp2) folded into to_q before the one
                # rounding to bf16, so the flash kernel's exponent needs no multiply (ccnet_attention_d64_prescaled_bf16)
                "qkv1s": plin(torch.cat([b.attn1.to_q.weight.detach().float() * qscale, b.attn1.to_k.weight.detach().float(),
                                         b.attn1.to_v.weight.detach().float()], 0)) if ATTN_PRESCALE and a.head_ch == 64 else None,
                "q2s": plin(b.attn2.to_q.weight.detach().float() * qscale) if ATTN_PRESCALE and a.head_ch == 64 else None,
                "o1": (plin(b.attn1.to_out[0].weight), f32(b.attn1.to_out[0].bias)),
                "q2": plin(b.attn2.to_q.weight),
                "kv2": plin(torch.cat([b.attn2.to_k.weight, b.attn2.to_v.weight], 0)),
                "o2": (plin(b.attn2.to_out[0].weight), f32(b.attn2.to_out[0].bias)),
                "ff1": (plin(ops.geglu_interleave(b.ff.linear1.weight.detach())),
                        f32(ops.geglu_interleave(b.ff.linear1.bias.detach()))),
                "ff2": (plin(b.ff.linear2.weight), f32(b.ff.linear2.bias)),
            }

        pk["conv_in"] = (f32(self.conv_in.conv.weight), f32(self.conv_in.conv.bias))
        pk["te0"] = (f32(self.time_embed[0].weight), f32(self.time_embed[0].bias))
        pk["te2"] = (f32(self.time_embed[2].weight), f32(self.time_embed[2].bias))
        c0 = self.block_out_channels[0]
        half = c0 // 2
        pk["freqs"] = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
        if self.num_class_embeds is not None:
            ce = self.class_embedding
            if isinstance(ce, nn.Embedding):
                pk["class_table"] = f32(ce.weight)
            else:
                pk["ce0"] = (f32(ce[0].weight), f32(ce[0].bias))
                pk["ce2"] = (f32(ce[2].weight), f32(ce[2].bias))
        # all time_emb_proj layers as one fp32 matrix -> one launch per forward
        rs = self._resnets_in_order()
        pk["tproj_w"] = torch.cat([f32(r.time_emb_proj.weight) for r in rs], 0).contiguous()
        pk["tproj_b"] = torch.cat([f32(r.time_emb_proj.bias) for r in rs], 0).contiguous()
        offs, o = [], 0
        for r in rs:
            offs.append(o)
            o += r.cout
        pk["tproj_off"] = {id(r): off for r, off in zip(rs, offs)}
        pk["res"] = {id(r): pres(r) for r in rs}
        atts = []
        for blk in self.down_blocks:
            if hasattr(blk, "attentions"):
                atts.extend(blk.attentions)
        atts.append(self.middle_block.attention)
        if hasattr(self, "up_blocks"):
            for blk in self.up_blocks:
                if hasattr(blk, "attentions"):
                    atts.extend(blk.attentions)
        pk["att"] = {id(a): pattn(a) for a in atts}
        pk["att_order"] = atts
        pk["down"] = {id(b): pconv(b.downsampler.op.conv) for b in self.down_blocks if b.downsampler is not None}
        self._pack_extra(pk, pconv, f32)
        self._pk = pk
        self._pk_gen = getattr(self, "_pk_gen", 0) + 1  # samplers key their captured graphs on it (sampling._graph_for)
        return pk

    def _pack_extra(self, pk, pconv, f32):
        pass

    # ---- lowering helpers
    def _embedding(self, pk, timesteps, class_labels, B):
        dev = pk["freqs"].device
        t = timesteps.to(device=dev, dtype=torch.float32).reshape(-1)
        if t.numel() not in (1, B):
            raise ValueError(f"timesteps must have 1 or {B} elements, got {t.numel()}")
        c0 = self.block_out_channels[0]
        t_emb = ops.timestep_embedding(t, B, c0, pk["freqs"])
        h = ops.linear_f32(t_emb, pk["te0"][0], pk["te0"][1], silu_out=True)
        add = None
        if self.num_class_embeds is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            if "class_table" in pk:
                add = pk["class_table"][class_labels.to(dev).long().reshape(-1)]
            else:
                cl = class_labels.to(device=dev, dtype=torch.float32).reshape(B, 1).contiguous()
                c1 = ops.linear_f32(cl, pk["ce0"][0], pk["ce0"][1], silu_out=True)
                add = ops.linear_f32(c1, pk["ce2"][0], pk["ce2"][1])
        # emb = time_embed(t_emb) + class_emb is only ever consumed as silu(emb) (ResnetBlock.time_emb_proj input),
        # so the SiLU is applied once here instead of once per output row of the projection below
        semb = ops.linear_f32(h, pk["te2"][0], pk["te2"][1], add=add, silu_out=2)
        # every resnet's time_emb_proj(silu(emb)) in one launch; resnets take column slices
        return ops.linear_f32(semb, pk["tproj_w"], pk["tproj_b"])

    @torch.no_grad()
    def embedding_table(self, timesteps, class_labels, B, max_bytes: int = 1 << 30):
        """Every step's time-embedding projections at once: [T, B, sum of resnet couts] fp32 (row t = what _embedding
        returns for timesteps[t]), or None when the table would exceed `max_bytes`.  The class-embedding MLP and the
        three time-embedding linears depend on (t, image) only, so a sampler evaluates them once per batch (weights read
        once instead of once per step: at one image that is 12 of a step's launches and ~100 MB of fp32 weights per network)."""
        pk = self._pk or self._pack()
        dev = pk["freqs"].device
        t = torch.as_tensor([float(v) for v in timesteps], dtype=torch.float32, device=dev)
        T = t.numel()
        if T * B * pk["tproj_w"].shape[0] * 4 > max_bytes:
            return None
        c0 = self.block_out_channels[0]
        t_emb = ops.timestep_embedding(t, T, c0, pk["freqs"])
        h = ops.linear_f32(t_emb, pk["te0"][0], pk["te0"][1], silu_out=True)
        add = None
        if self.num_class_embeds is not None:
            if class_labels is None:
                raise ValueError("class_labels should be provided when num_class_embeds > 0")
            if "class_table" in pk:
                add = pk["class_table"][class_labels.to(dev).long().reshape(-1)]
            else:
                cl = class_labels.to(device=dev, dtype=torch.float32).reshape(B, 1).contiguous()
                c1 = ops.linear_f32(cl, pk["ce0"][0], pk["ce0"][1], silu_out=True)
                add = ops.linear_f32(c1, pk["ce2"][0], pk["ce2"][1])
            add = add.repeat(T, 1)  # row t*B + b
        semb = ops.linear_f32(h.repeat_interleave(B, 0), pk["te2"][0], pk["te2"][1], add=add, silu_out=2)
        return ops.linear_f32(semb, pk["tproj_w"], pk["tproj_b"]).view(T, B, -1)

    def _context(self, context, B):
        if context is None:
            raise ValueError("context is required (with_conditioning=True)")
        if context.shape[0] != B:
            raise ValueError(f"context batch {context.shape[0]} != input batch {B}")
        S, D = context.shape[1], context.shape[2]
        if D % 64:
            raise CcnetError("cross_attention_dim must be a multiple of 64 for the MI355X lowering")
        c = context.reshape(B * S, D, 1, 1)
        a = ops.nchw_to_act(c, D)  # fp32 -> bf16 rows
        return a.t, S

    def context_kv(self, context):
        """Cross-attention K/V of `context` for every SpatialTransformer, in execution order.  They do not depend
        on the timestep, so a sampler may compute them once per batch and pass them back via `_ctx_kv`."""
        pk = self._pk or self._pack()
        ctx, S = self._context(context, context.shape[0])
        out = []
        for a in pk["att_order"]:
            p = pk["att"][id(a)]
            kv = ops.linear(ctx, p["kv2"], None, 2 * p["inner"], ctx.shape[1])
            if p["xap"] is not None:  # the same keys / values as the LDS images of the fused cross-attention kernel
                kv._ximg = ops.xattn_kv_image(kv, kv.shape[0] // S, S, p["inner"])
            out.append(kv)
        return out, S

    def _run_resnet(self, pk, r, x: Act, tall, x2: Optional[Act] = None, res2: Optional[Act] = None):
        p = pk["res"][id(r)]
        g, eps = self.norm_num_groups, self.norm_eps
        off = pk["tproj_off"][id(r)]
        cout = p["cout"]
        h = ops.groupnorm(x, p["n1"][0], p["n1"][1], g, eps, True, x2=x2)
        h = ops.conv(h, p["c1"][0], p["c1"][1], cout, R=3, rowvec=tall[:x.N, off:off + cout])
        h = ops.groupnorm(h, p["n2"][0], p["n2"][1], g, eps, True)
        if p["skip"] is None:
            skip = x
        else:
            skip = ops.conv(x, p["skip"][0], p["skip"][1], cout, R=1, pad=0, x2=x2)
        return ops.conv(h, p["c2"][0], p["c2"][1], cout, R=3, res1=skip, res2=res2)

    @staticmethod
    def _cross_attn(p, t, x: Act, kv, S, B, T, heads, hc, inner, scale, pre, inl, dup_to, attend):
        """t + attn2(norm2(t), context) of a transformer block (rows t, the block's input x).  Returns (t, halves, x, B): with shared
        CFG halves (dup_to) the result has dup_to images while `halves` keeps the B-image block input for proj_out's residual."""
        ximg = getattr(kv, "_ximg", None) if p["xap"] is not None else None
        if ximg is not None and T % 128 == 0 and t.shape[0] >= FF_FUSED_MIN_ROWS and (dup_to is None or hc == 64):
            # norm2 -> to_q -> cross-attention -> to_out -> + t in one launch; with shared CFG halves once per half (the same B
            # query / residual rows, that half's keys and values) into the halves of one tensor
            try:
                if dup_to is None:
                    return ops.xattn_fused(t, inner, heads, p["xap"], p["o2"][1], p["ln2"][0], p["ln2"][1], ximg, S, T), None, x, B
                t2 = torch.empty((2 * B * T, inner), dtype=t.dtype, device=t.device)
                for hf in (0, 1):
                    ops.xattn_fused(t, inner, heads, p["xap"], p["o2"][1], p["ln2"][0], p["ln2"][1], ximg[hf * B:(hf + 1) * B], S, T,
                                    out=t2[hf * B * T:(hf + 1) * B * T])
                return t2, x, x, B
            except CcnetUnsupported:  # (decided from the shapes before anything is launched)
                pass
        q = None
        if inl:
            w, b, cs = p["q2i"]
            try:
                q = ops.linear(t, w, b, inner, inner, colsum=cs)
            except CcnetUnsupported:
                pass
        if q is None:
            q = ops.linear(ops.layernorm(t, inner, *p["ln2"]), p["q2s"] if pre else p["q2"], None, inner, inner)
        if dup_to is not None and hc == 64:
            # From here on the two halves differ (their prompts do).  Nothing is copied: the cross-attention and its output
            # projection run once per half -- the same B query / residual rows, that half's keys and values -- into the
            # two halves of one tensor, and so does proj_out in the caller.
            o = torch.empty((2 * B * T, inner), dtype=t.dtype, device=t.device)
            t2 = torch.empty((2 * B * T, inner), dtype=t.dtype, device=t.device)
            rows = lambda tt, hf, n: Act(tt[hf * B * n:(hf + 1) * B * n], 1, B * n, 1, tt.shape[1])
            for hf in (0, 1):
                kvh = kv[hf * B * S:(hf + 1) * B * S]
                ops.attention_d64(q, kvh[:, :inner], kvh[:, inner:], B, heads, T, S, scale, prescaled=pre, out=o[hf * B * T:(hf + 1) * B * T])
                ops.conv(rows(o, hf, T), p["o2"][0], p["o2"][1], inner, R=1, pad=0, res1=Act(t, 1, B * T, 1, inner), out=rows(t2, hf, T))
            return t2, x, x, B
        if dup_to is not None:
            q, t, x, B = torch.cat([q, q]), torch.cat([t, t]), _dup_act(x), dup_to
        o = attend(q, kv[:, :inner], kv[:, inner:], S, B)
        return ops.linear(o, p["o2"][0], p["o2"][1], inner, inner, res1=t), None, x, B

    def _run_attn(self, pk, a, x: Act, kv: torch.Tensor, S: int, res2: Optional[Act] = None, dup_to: Optional[int] = None):
        """dup_to: x holds the first x.N images of a batch of dup_to = 2 x.N whose halves are identical so far (CFG_SHARE): the
        block runs on x.N images up to attn2's query and on dup_to images from the cross-attention on (kv, res2 and the result
        have dup_to images)."""
        if dup_to is not None and LN_FOLD:
            x, dup_to = _dup_act(x), None
        p = pk["att"][id(a)]
        C_, inner, heads, hc = x.C, p["inner"], p["heads"], p["head_ch"]
        B, T = x.N, x.H * x.W
        scale = 1.0 / math.sqrt(hc)
        if hc % 64:
            raise CcnetError(f"attention head dim {hc}: the MI355X lowering needs num_head_channels % 64 == 0")

        pre = (not LN_FOLD) and p["qkv1s"] is not None
        halves = None

        def attend(q, k, v, Skv, Bq=None):
            Bq = B if Bq is None else Bq
            if hc == 64:  # flash kernel (SD-2.1 geometry)
                return ops.attention_d64(q, k, v, Bq, heads, T, Skv, scale, prescaled=pre)
            return ops.attention_generic(q, k, v, Bq, heads, T, Skv, hc, scale)  # e.g. one head of 512 / 768 (repo YAML)
        h = ops.groupnorm(x, p["norm"][0], p["norm"][1], self.norm_num_groups, self.transformer_norm_eps, False)
        if LN_FOLD:
            # The three LayerNorms never materialise.  Their row statistics come out of the epilogue of the layer that writes
            # the rows (per-span partials, ops.conv(emit_stats=True); a plan that cannot emit them falls back to a statistics
            # pass), and the consuming GEMM applies them to the raw rows (folded gamma / beta, ops.fold_layernorm).
            stats_of = lambda tt, st: st if st is not None else ops.row_stats(tt, inner)
            ta, st = ops.conv(h, p["pin"][0], p["pin"][1], inner, R=1, pad=0, emit_stats=True)
            t = ta.t
            w, b, cs = p["qkv1f"]
            qkv = ops.linear(t, w, b, 3 * inner, inner, rowstat=stats_of(t, st), colsum=cs)
            o = attend(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], T)
            t, st = ops.linear(o, p["o1"][0], p["o1"][1], inner, inner, res1=t, emit_stats=True)
            w, b, cs = p["q2f"]
            q = ops.linear(t, w, b, inner, inner, rowstat=stats_of(t, st), colsum=cs)
            o = attend(q, kv[:, :inner], kv[:, inner:], S)
            t, st = ops.linear(o, p["o2"][0], p["o2"][1], inner, inner, res1=t, emit_stats=True)
            w, b, cs = p["ff1f"]  # norm3 + linear1 + GEGLU in one kernel
            gl = ops.linear(t, w, b, 8 * inner, inner, act=2, rowstat=stats_of(t, st), colsum=cs)
        else:
            t = ops.conv(h, p["pin"][0], p["pin"][1], inner, R=1, pad=0).t
            inl = p["qkv1i"] is not None and ops.ln_inline_ok(t.shape[0], inner, inner)
            qkv = None
            if inl:
                # (the library has the last word on which launches take their LayerNorm statistics themselves -- tuning switches
                # and descriptor limits also decide: a declined launch runs as LayerNorm kernel + plain layer)
                w, b, cs = p["qkv1i"]
                try:
                    qkv = ops.linear(t, w, b, 3 * inner, inner, colsum=cs)
                except CcnetUnsupported:
                    inl = False
            if qkv is None:
                qkv = ops.linear(ops.layernorm(t, inner, *p["ln1"]), p["qkv1s"] if pre else p["qkv1"], None, 3 * inner, inner)
            o = attend(qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:], T)
            t = ops.linear(o, p["o1"][0], p["o1"][1], inner, inner, res1=t)
            t, halves, x, B = self._cross_attn(p, t, x, kv, S, B, T, heads, hc, inner, scale, pre, inl, dup_to, attend)
            if p["ffp"] is not None and t.shape[0] >= FF_FUSED_MIN_ROWS:
                gl = None
                t = ops.ff_fused(t, inner, p["ffp"], p["ff2"][1], p["ln3"][0], p["ln3"][1])
            else:
                gl = ops.linear(ops.layernorm(t, inner, *p["ln3"]), p["ff1"][0], p["ff1"][1], 8 * inner, inner, act=2)
        if gl is not None:
            t = ops.linear(gl, p["ff2"][0], p["ff2"][1], inner, 4 * inner, res1=t)
        if not LN_FOLD and halves is not None:
            # proj_out + the block's input (B rows, shared by the halves) [+ the ControlNet residual of each half]
            Bh, HW = halves.N, halves.H * halves.W
            full = ops.new_act(2 * Bh, halves.H, halves.W, C_, t.device)
            full2 = ops.new_act(2 * Bh, halves.H, halves.W, C_, t.device) if res2 is not None else None
            part = lambda a_, hf: Act(a_.t[hf * Bh * HW:(hf + 1) * Bh * HW], Bh, a_.H, a_.W, a_.C)
            for hf in (0, 1):
                ops.conv(Act(t[hf * Bh * HW:(hf + 1) * Bh * HW], Bh, halves.H, halves.W, inner), p["pout"][0], p["pout"][1], C_, R=1,
                         pad=0, res1=halves, res2=part(res2, hf) if res2 is not None else None, out=part(full, hf),
                         out2=part(full2, hf) if res2 is not None else None)
            return (full, full2) if res2 is not None else full
        ta = Act(t, x.N, x.H, x.W, inner)
        return ops.conv(ta, p["pout"][0], p["pout"][1], C_, R=1, pad=0, res1=x, res2=res2)

    def _encode(self, pk, h: Act, tall, kvs, S, add: Optional[Sequence[Act]] = None, dup_to: Optional[int] = None):
        """Down blocks. Returns (h, skips); when `add` is given skips[k] = s_k + add[k] (written by the same launch
        that produces s_k) while the trunk continues with the un-added s_k, as upstream does.  dup_to: see _run_attn -- h
        holds half the batch; the first resnet and the head of the first transformer block run on it (only asked for when the
        first level has attention)."""
        skips: List[Act] = []
        k = 1  # skip 0 is the conv_in output, handled by the caller
        ai = 0

        def pick(out):
            nonlocal k
            if add is not None:
                o, o2 = out
                skips.append(o2)
            else:
                o = out
                skips.append(o)
            k += 1
            return o

        for blk in self.down_blocks:
            has_att = hasattr(blk, "attentions")
            for j, r in enumerate(blk.resnets):
                r2 = add[k] if (add is not None) else None
                if has_att:
                    h = self._run_resnet(pk, r, h, tall)
                    h = pick(self._run_attn(pk, blk.attentions[j], h, kvs[ai], S, res2=r2, dup_to=dup_to))
                    dup_to = None
                    ai += 1
                else:
                    h = pick(self._run_resnet(pk, r, h, tall, res2=r2))
            if blk.downsampler is not None:
                w, b = pk["down"][id(blk)]
                r2 = add[k] if (add is not None) else None
                h = pick(ops.conv(h, w, b, h.C, R=3, stride=2, res2=r2))
        return h, skips, ai

    def _share_ok(self, asked: bool, Bx: int, B: int) -> bool:
        """Whether the CFG halves share their head (CFG_SHARE): the caller vouches for equal inputs, x holds one half, and the
        first level has a transformer block to duplicate in."""
        return bool(asked) and CFG_SHARE and B == 2 * Bx and hasattr(self.down_blocks[0], "attentions")

    def _mid(self, pk, h, tall, kv, S, res2=None):
        m = self.middle_block
        h = self._run_resnet(pk, m.resnet_1, h, tall)
        h = self._run_attn(pk, m.attention, h, kv, S)
        return self._run_resnet(pk, m.resnet_2, h, tall, res2=res2)


def _dup_act(a: Act) -> Act:
    """[a; a] along the batch: the two halves of a CFG batch that are identical so far (CFG_SHARE)."""
    return Act(torch.cat([a.t, a.t]), 2 * a.N, a.H, a.W, a.C)


def _as_act(t, N, dev) -> Act:
    """Accept a residual as Act, as a zero-copy channels-last bf16 NCHW view, or as any NCHW tensor."""
    if isinstance(t, Act):
        return t
    if t.dim() != 4:
        raise ValueError("residual must be NCHW")
    n, c, h, w = t.shape
    if t.dtype == BF16 and t.is_cuda and t.stride(1) == 1 and t.stride(3) == t.stride(0) // (h * w) \
            and t.stride(2) == w * t.stride(3):
        ld = t.stride(3)
        base = torch.as_strided(t, (n * h * w, c), (ld, 1), t.storage_offset())
        return Act(base, n, h, w, c)
    return ops.nchw_to_act(t.to(dev), c)


def _act_view_nchw(a: Act) -> torch.Tensor:
    """Zero-copy NCHW-shaped (channels-last strided) bf16 view of an activation."""
    return a.t.view(a.N, a.H, a.W, a.t.shape[1])[..., : a.C].permute(0, 3, 1, 2)


class DiffusionModelUNet(_Trunk):
    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, num_res_blocks=(2, 2, 2, 2),
                 num_channels=(32, 64, 64, 64), attention_levels=(False, False, True, True),
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, resblock_updown: bool = False,
                 num_head_channels=8, with_conditioning: bool = False, transformer_num_layers: int = 1,
                 cross_attention_dim: Optional[int] = None, num_class_embeds: Optional[int] = None,
                 upcast_attention: bool = False, use_flash_attention: bool = False):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("only spatial_dims=2 is on the ccnet hot path")
        if resblock_updown:
            raise NotImplementedError("resblock_updown=True")
        if in_channels > 8 or out_channels > 8:
            raise CcnetError("in/out channels > 8 are not lowered (latent channels are 3 or 4 in every ccnet config)")
        ted = self._build_trunk(in_channels, num_res_blocks, num_channels, attention_levels, norm_num_groups,
                                norm_eps, num_head_channels, with_conditioning, transformer_num_layers,
                                cross_attention_dim, num_class_embeds)
        self.out_channels = out_channels
        n = len(num_channels)
        rev_ch = list(reversed(num_channels))
        rev_res = list(reversed(self.num_res_blocks_))
        rev_att = list(reversed(self.attention_levels_))
        rev_hc = list(reversed(self.num_head_channels_))
        self.up_blocks = nn.ModuleList()
        out_ch = rev_ch[0]
        for i in range(n):
            prev, out_ch = out_ch, rev_ch[i]
            in_ch = rev_ch[min(i + 1, n - 1)]
            blk = _Holder()
            blk.resnets = nn.ModuleList()
            att = bool(rev_att[i]) and with_conditioning
            atts = []
            nres = rev_res[i] + 1
            for j in range(nres):
                skip = in_ch if j == nres - 1 else out_ch
                rin = prev if j == 0 else out_ch
                blk.resnets.append(_resnet(rin + skip, ted, out_ch, norm_num_groups, norm_eps))
                if att:
                    atts.append(_spatial_transformer(out_ch, out_ch // rev_hc[i], rev_hc[i], 1, norm_num_groups,
                                                     norm_eps, cross_attention_dim))
            if att:
                blk.attentions = nn.ModuleList(atts)
            if i != n - 1:
                us = _Holder()
                us.conv = _convolution(out_ch, out_ch)
                blk.upsampler = us
            else:
                blk.upsampler = None
            self.up_blocks.append(blk)
        self.out = nn.Sequential(nn.GroupNorm(norm_num_groups, num_channels[0], eps=norm_eps, affine=True), nn.SiLU(),
                                 _zero(_convolution(num_channels[0], out_channels)))

    def _pack_extra(self, pk, pconv, f32):
        pk["up"] = {id(b): pconv(b.upsampler.conv.conv) for b in self.up_blocks if b.upsampler is not None}
        pk["out_norm"] = (f32(self.out[0].weight), f32(self.out[0].bias))
        w = self.out[2].conv.weight.detach().float()  # [Cout, C0, 3, 3] -> bf16 [Cout][9][C0]
        dev = pk["freqs"].device
        pk["out_conv"] = (w.permute(0, 2, 3, 1).reshape(w.shape[0], 9, w.shape[1]).to(BF16).contiguous().to(dev),
                          f32(self.out[2].conv.bias))

    @torch.no_grad()
    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, context: Optional[torch.Tensor] = None,
                class_labels: Optional[torch.Tensor] = None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, _ctx_kv=None, _x_batch: Optional[int] = None, _tall=None,
                _cfg_shared: bool = False):
        """x fp32 NCHW [Bx, Cin, h, w]; the network batch is `_x_batch or Bx` (image n reads x[n % Bx], which lets a
        CFG sampler pass the un-duplicated latent).  _cfg_shared: the caller guarantees that images n and n + Bx also share
        class label and timestep (CFG_SHARE above).  Returns fp32 NCHW [B, Cout, h, w]."""
        pk = self._pk or self._pack()
        dev = pk["freqs"].device
        if not x.is_cuda:
            raise CcnetError("ccnet_amd.DiffusionModelUNet runs on the GPU only (no CPU fallback)")
        x = x.contiguous().float()
        B = _x_batch or x.shape[0]
        tall = _tall if _tall is not None else self._embedding(pk, timesteps, class_labels, B)
        if _ctx_kv is None:
            kvs, S = self.context_kv(context)
        else:
            kvs, S = _ctx_kv
        c0 = self.block_out_channels[0]
        add = None
        if down_block_additional_residuals is not None:
            add = [_as_act(t, B, dev) for t in down_block_additional_residuals]
        w, b = pk["conv_in"]
        dup_to = B if self._share_ok(_cfg_shared, x.shape[0], B) else None
        h = ops.conv_in_small(x, x.shape[0] if dup_to else B, w, b, c0, c0)
        if add is not None:
            s0 = ops.conv_in_small(x, B, w, b, c0, c0, res=add[0])
        else:
            s0 = _dup_act(h) if dup_to else h
        h, skips, ai = self._encode(pk, h, tall, kvs, S, add, dup_to=dup_to)
        skips = [s0] + skips
        if add is not None and len(add) != len(skips):
            raise ValueError(f"expected {len(skips)} down_block_additional_residuals, got {len(add)}")
        mid_add = _as_act(mid_block_additional_residual, B, dev) if mid_block_additional_residual is not None else None
        h = self._mid(pk, h, tall, kvs[ai], S, res2=mid_add)
        if mid_add is not None:
            h = h[1]
        ai += 1
        return self._decode(pk, h, skips, tall, kvs, S, ai)

    def _decode(self, pk, h, skips, tall, kvs, S, ai):
        """Up blocks (each resnet reads its skip as the second source of a channel concat) and the output head."""
        skips = list(skips)
        for blk in self.up_blocks:
            has_att = hasattr(blk, "attentions")
            for j, r in enumerate(blk.resnets):
                s = skips.pop()
                h = self._run_resnet(pk, r, h, tall, x2=s)
                if has_att:
                    h = self._run_attn(pk, blk.attentions[j], h, kvs[ai], S)
                    ai += 1
            if blk.upsampler is not None:
                uw, ub = pk["up"][id(blk)]
                h = ops.conv(h, uw, ub, h.C, R=3, ups=2)
        h = ops.groupnorm(h, pk["out_norm"][0], pk["out_norm"][1], self.norm_num_groups, self.norm_eps, True)
        return ops.conv_out_small(h, pk["out_conv"][0], pk["out_conv"][1], self.out_channels)

    # ---- two-phase form for a sampler that overlaps this network's encoder with the ControlNet (sampling.Sampler):
    # upstream's trunk never sees the ControlNet residuals (they are added to the skip copies and to the middle block's
    # output only), so encoder + middle block depend on (x, t) alone and can run beside the ControlNet; the ControlNet's
    # zero-convolutions then write skip + residual directly (their res1 operand), and decode_phase() continues from there.
    @torch.no_grad()
    def encode_phase(self, x: torch.Tensor, timesteps: torch.Tensor, context: Optional[torch.Tensor] = None,
                     class_labels: Optional[torch.Tensor] = None, _ctx_kv=None, _x_batch: Optional[int] = None,
                     _tall=None, _cfg_shared: bool = False):
        pk = self._pk or self._pack()
        if not x.is_cuda:
            raise CcnetError("ccnet_amd.DiffusionModelUNet runs on the GPU only (no CPU fallback)")
        x = x.contiguous().float()
        B = _x_batch or x.shape[0]
        tall = _tall if _tall is not None else self._embedding(pk, timesteps, class_labels, B)
        kvs, S = self.context_kv(context) if _ctx_kv is None else _ctx_kv
        c0 = self.block_out_channels[0]
        w, b = pk["conv_in"]
        dup_to = B if self._share_ok(_cfg_shared, x.shape[0], B) else None
        s0 = ops.conv_in_small(x, x.shape[0] if dup_to else B, w, b, c0, c0)
        h, skips, ai = self._encode(pk, s0, tall, kvs, S, None, dup_to=dup_to)
        h = self._mid(pk, h, tall, kvs[ai], S)
        # (shared CFG head: skips[0] stays at B images -- the ControlNet's first zero-convolution adds its own half-batch
        # residual to it and duplicates the sum; decode_phase duplicates it if no ControlNet did)
        return {"pk": pk, "tall": tall, "kvs": kvs, "S": S, "ai": ai + 1, "skips": [s0] + skips, "h": h, "B": B}

    @torch.no_grad()
    def decode_phase(self, st, skips: Optional[Sequence[Act]] = None, h: Optional[Act] = None) -> torch.Tensor:
        """Decoder over `skips` / `h` (default: the encoder's own, i.e. no ControlNet).  Returns fp32 NCHW."""
        skips = st["skips"] if skips is None else skips
        if len(skips) != len(st["skips"]):
            raise ValueError(f"expected {len(st['skips'])} skips, got {len(skips)}")
        if 2 * skips[0].N == st.get("B", skips[0].N):  # shared CFG head: the first skip is still one half
            skips = [_dup_act(skips[0])] + list(skips[1:])
        return self._decode(st["pk"], st["h"] if h is None else h, skips, st["tall"], st["kvs"], st["S"], st["ai"])


class ControlNet(_Trunk):
    def __init__(self, spatial_dims: int, in_channels: int, num_res_blocks=(2, 2, 2, 2),
                 num_channels=(32, 64, 64, 64), attention_levels=(False, False, True, True),
                 norm_num_groups: int = 32, norm_eps: float = 1e-6, resblock_updown: bool = False,
                 num_head_channels=8, with_conditioning: bool = False, transformer_num_layers: int = 1,
                 cross_attention_dim: Optional[int] = None, num_class_embeds: Optional[int] = None,
                 upcast_attention: bool = False, use_flash_attention: bool = False,
                 conditioning_embedding_in_channels: int = 1,
                 conditioning_embedding_num_channels: Sequence[int] = (16, 32, 96, 256)):
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("only spatial_dims=2 is on the ccnet hot path")
        if resblock_updown:
            raise NotImplementedError("resblock_updown=True")
        if in_channels > 8 or conditioning_embedding_in_channels > 8:
            raise CcnetError("in_channels / conditioning_embedding_in_channels > 8 are not lowered")
        self._build_trunk(in_channels, num_res_blocks, num_channels, attention_levels, norm_num_groups, norm_eps,
                          num_head_channels, with_conditioning, transformer_num_layers, cross_attention_dim,
                          num_class_embeds)
        ce = _Holder()
        ech = tuple(conditioning_embedding_num_channels)
        ce.conv_in = _convolution(conditioning_embedding_in_channels, ech[0])
        ce.blocks = nn.ModuleList()
        for i in range(len(ech) - 1):
            ce.blocks.append(_convolution(ech[i], ech[i]))
            ce.blocks.append(_convolution(ech[i], ech[i + 1], 3, 2, 1))
        ce.conv_out = _zero(_convolution(ech[-1], num_channels[0]))
        self.controlnet_cond_embedding = ce
        self.cond_channels = ech
        n = len(num_channels)
        self.controlnet_down_blocks = nn.ModuleList()
        out_ch = num_channels[0]
        self.controlnet_down_blocks.append(_zero(nn.Conv2d(out_ch, out_ch, 1)))
        for i in range(n):
            out_ch = num_channels[i]
            for _ in range(self.num_res_blocks_[i]):
                self.controlnet_down_blocks.append(_zero(nn.Conv2d(out_ch, out_ch, 1)))
            if i != n - 1:
                self.controlnet_down_blocks.append(_zero(nn.Conv2d(out_ch, out_ch, 1)))
        self.controlnet_mid_block = _zero(nn.Conv2d(num_channels[-1], num_channels[-1], 1))

    def _pack_extra(self, pk, pconv, f32):
        ce = self.controlnet_cond_embedding
        pk["ce_in"] = (f32(ce.conv_in.conv.weight), f32(ce.conv_in.conv.bias))
        blocks = []
        for b in ce.blocks:
            ci, co = b.conv.in_channels, b.conv.out_channels
            blocks.append(pconv(b.conv, _pad64(ci), _pad64(co)) + (b.conv.stride[0], _pad64(co)))
        pk["ce_blocks"] = blocks
        pk["ce_out"] = pconv(ce.conv_out.conv, _pad64(ce.conv_out.conv.in_channels), None)
        pk["zero"] = [pconv(c) for c in self.controlnet_down_blocks]
        pk["zero_mid"] = pconv(self.controlnet_mid_block)

    @torch.no_grad()
    def cond_embedding(self, controlnet_cond: torch.Tensor, _max_bytes: int = 2 ** 31 - 1) -> Act:
        """controlnet_cond_embedding(cond): step-invariant, so a sampler computes it once per image."""
        pk = self._pk or self._pack()
        c = controlnet_cond.contiguous().float()
        if not c.is_cuda:
            raise CcnetError("ccnet_amd.ControlNet runs on the GPU only (no CPU fallback)")
        e0 = self.cond_channels[0]
        # the first activations live at the conditioning image's resolution: keep each below the conv kernel's 2 GiB limit
        per_image = c.shape[2] * c.shape[3] * max(_pad64(ch) for ch in self.cond_channels) * 2
        chunk = _even_chunk(c.shape[0], max(1, _max_bytes // per_image))
        if c.shape[0] > chunk:
            parts = [self.cond_embedding(c[i:i + chunk], _max_bytes) for i in range(0, c.shape[0], chunk)]
            return Act(torch.cat([p_.t for p_ in parts]), c.shape[0], parts[0].H, parts[0].W, parts[0].C)
        h = ops.conv_in_small(c, c.shape[0], pk["ce_in"][0], pk["ce_in"][1], e0, _pad64(e0), act=1)
        for w, b, stride, cop in pk["ce_blocks"]:
            h = ops.conv(h, w, b, cop, R=3, stride=stride, act=1)
        w, b = pk["ce_out"]
        return ops.conv(h, w, b, self.block_out_channels[0], R=3)

    @torch.no_grad()
    def forward_native(self, x, timesteps, controlnet_cond, conditioning_scale: float = 1.0, context=None,
                       class_labels=None, _ctx_kv=None, _cond_emb: Optional[Act] = None,
                       _x_batch: Optional[int] = None, _unet_state=None, _before_zero=None, _tall=None,
                       _cfg_shared: bool = False):
        """Same computation as forward(); returns the residuals as NHWC bf16 `Act`s.  With `_unet_state` (the result of
        DiffusionModelUNet.encode_phase on the same inputs) the zero-convolutions add the UNet's skips / middle output
        in their epilogue and the returned tensors are skip + residual, ready for decode_phase(); `_before_zero()` is
        called after the trunk and before those launches (the sampler's stream join)."""
        pk = self._pk or self._pack()
        if not x.is_cuda:
            raise CcnetError("ccnet_amd.ControlNet runs on the GPU only (no CPU fallback)")
        x = x.contiguous().float()
        B = _x_batch or x.shape[0]
        tall = _tall if _tall is not None else self._embedding(pk, timesteps, class_labels, B)
        if _ctx_kv is None:
            kvs, S = self.context_kv(context)
        else:
            kvs, S = _ctx_kv
        ce = _cond_emb if _cond_emb is not None else self.cond_embedding(controlnet_cond)
        c0 = self.block_out_channels[0]
        w, b = pk["conv_in"]
        dup_to = B if self._share_ok(_cfg_shared, x.shape[0], B) else None
        h = ops.conv_in_small(x, x.shape[0] if dup_to else B, w, b, c0, c0, res=ce)  # h = conv_in(x) + cond_emb (image n uses ce[n % ce.N])
        s0 = h  # (shared CFG head: B images; its zero-convolution below runs on them and the result is duplicated)
        h, skips, ai = self._encode(pk, h, tall, kvs, S, dup_to=dup_to)
        skips = [s0] + skips
        h = self._mid(pk, h, tall, kvs[ai], S)
        sc = float(conditioning_scale)
        if _before_zero is not None:
            _before_zero()
        us = _unet_state["skips"] if _unet_state is not None else [None] * len(skips)
        if len(us) != len(skips):
            raise ValueError(f"UNet state holds {len(us)} skips, the ControlNet produces {len(skips)}")
        if dup_to and us[0] is not None and us[0].N != skips[0].N:
            us = [Act(us[0].t[:skips[0].rows], skips[0].N, us[0].H, us[0].W, us[0].C)] + list(us[1:])  # (a full-batch UNet skip: its first half)
        down = [ops.conv(s, zw, zb, s.C, R=1, pad=0, scale=sc, res1=u) for s, (zw, zb), u in zip(skips, pk["zero"], us)]
        if dup_to:
            down[0] = _dup_act(down[0])
        zw, zb = pk["zero_mid"]
        mid = ops.conv(h, zw, zb, h.C, R=1, pad=0, scale=sc, res1=_unet_state["h"] if _unet_state is not None else None)
        return down, mid

    def forward(self, x, timesteps, controlnet_cond, conditioning_scale: float = 1.0, context=None,
                class_labels=None):
        """Returns (list of NCHW-shaped bf16 tensors, mid) -- channels-last views of the NHWC activations, which
        DiffusionModelUNet.forward consumes without a copy."""
        down, mid = self.forward_native(x, timesteps, controlnet_cond, conditioning_scale, context, class_labels)
        return [_act_view_nchw(d) for d in down], _act_view_nchw(mid)


def adjust_class_embedding(model: nn.Module) -> None:
    """util.py:1094-1132: continuous acquisition-time conditioning -- Linear(1, ted) -> SiLU -> Linear(ted, ted)."""
    name = str(model.__class__.__name__)
    if not (("DiffusionModelUNet" in name or "ControlNet" in name) and hasattr(model, "block_out_channels")):
        if hasattr(model, "config") or "ControlNetModel" in name:  # diffusers naming convention (util.py:1113-1123)
            model.config.num_class_embeds = int(99999)
            ted = model.config.time_embedding_dim or model.config.block_out_channels[0] * 4
            dev = next(model.parameters()).device
            model.class_embedding = nn.Sequential(nn.Linear(1, ted), nn.SiLU(), nn.Linear(ted, ted)).to(dev)
            return
        raise ValueError(f"{name} has neither key ('x' AND 'block_out_channels') nor key 'config'. "
                         f"Current diffusion model keys: {model.__dict__.keys()}")
    model.num_class_embeds = int(99999)
    ted = model.block_out_channels[0] * 4
    dev = model.conv_in.conv.weight.device
    model.class_embedding = nn.Sequential(nn.Linear(1, ted), nn.SiLU(), nn.Linear(ted, ted)).to(dev)
    model._invalidate()


# --------------------------------------------------------------------------- AutoencoderKL (stage 1)
def _ae_resblock(cin, cout, groups, eps) -> _Holder:
    h = _Holder()
    h.norm1 = nn.GroupNorm(groups, cin, eps=eps, affine=True)
    h.conv1 = _convolution(cin, cout)
    h.norm2 = nn.GroupNorm(groups, cout, eps=eps, affine=True)
    h.conv2 = _convolution(cout, cout)
    h.nin_shortcut = _convolution(cin, cout, 1, 1, 0) if cin != cout else nn.Identity()
    h.kind, h.cin, h.cout = "res", cin, cout
    return h


def _ae_attention(ch, groups, eps) -> _Holder:
    """AttentionBlock of the autoencoder: one head over all channels (num_head_channels=None), Linear q/k/v/proj with
    bias, residual connection."""
    h = _Holder()
    h.norm = nn.GroupNorm(groups, ch, eps=eps, affine=True)
    h.to_q = nn.Linear(ch, ch)
    h.to_k = nn.Linear(ch, ch)
    h.to_v = nn.Linear(ch, ch)
    h.proj_attn = nn.Linear(ch, ch)
    h.kind, h.ch = "attn", ch
    return h


def _ae_sample(ch, kind) -> _Holder:
    h = _Holder()
    h.conv = _convolution(ch, ch, 3, 2 if kind == "down" else 1, 0 if kind == "down" else 1)
    h.kind = kind
    return h


class AutoencoderKL(nn.Module):
    """monai-generative AutoencoderKL lowered to the same conv / GroupNorm kernels."""

    def __init__(self, spatial_dims: int, in_channels: int = 1, out_channels: int = 1, num_res_blocks=(2, 2, 2, 2),
                 num_channels=(32, 64, 64, 64), attention_levels=(False, False, True, True),
                 latent_channels: int = 3, norm_num_groups: int = 32, norm_eps: float = 1e-6,
                 with_encoder_nonlocal_attn: bool = True, with_decoder_nonlocal_attn: bool = True,
                 use_flash_attention: bool = False, _decoder_num_res_blocks=None, _final_act: bool = False,
                 _encoder_out_channels: Optional[int] = None):
        """The three underscore options are not monai-generative's: they let `ccnet_amd.diffusers_dialect.AutoencoderKL`
        express the SD-VAE on the same block list (decoder levels with layers_per_block + 1 resnets, SiLU after the
        final GroupNorm of encoder and decoder, an encoder that emits 2 x latent_channels moments)."""
        super().__init__()
        if spatial_dims != 2:
            raise NotImplementedError("only spatial_dims=2")
        if in_channels > 8 or out_channels > 8 or latent_channels > 8 or (_encoder_out_channels or 0) > 8:
            raise CcnetError("in/out/latent channels > 8 are not lowered")
        if any(c % 64 for c in num_channels):
            raise CcnetError("the MI355X lowering needs every entry of num_channels to be a multiple of 64")
        n = len(num_channels)
        nrb = _tuple(num_res_blocks, n)
        att = _tuple(attention_levels, n)
        self.num_channels = list(num_channels)
        self.latent_channels, self.in_channels, self.out_channels = latent_channels, in_channels, out_channels
        self.norm_num_groups, self.norm_eps = norm_num_groups, norm_eps
        self.final_act = bool(_final_act)
        enc_out = _encoder_out_channels or latent_channels
        g, eps = norm_num_groups, norm_eps
        # encoder.blocks / decoder.blocks are flat lists upstream (keys `encoder.blocks.{i}....`)
        enc = [_convolution(in_channels, num_channels[0])]
        out_ch = num_channels[0]
        for i in range(n):
            in_ch, out_ch = out_ch, num_channels[i]
            for _ in range(nrb[i]):
                enc.append(_ae_resblock(in_ch, out_ch, g, eps))
                in_ch = out_ch
                if att[i]:
                    enc.append(_ae_attention(in_ch, g, eps))
            if i != n - 1:
                enc.append(_ae_sample(in_ch, "down"))
        if with_encoder_nonlocal_attn:
            cl = num_channels[-1]
            enc += [_ae_resblock(cl, cl, g, eps), _ae_attention(cl, g, eps), _ae_resblock(cl, cl, g, eps)]
        enc.append(nn.GroupNorm(g, num_channels[-1], eps=eps, affine=True))
        enc.append(_convolution(num_channels[-1], enc_out))
        self.encoder = _Holder()
        self.encoder.blocks = nn.ModuleList(enc)
        rev = list(reversed(num_channels))
        rev_res = list(reversed(_tuple(_decoder_num_res_blocks, n) if _decoder_num_res_blocks is not None else nrb))
        rev_att = list(reversed(att))
        dec = [_convolution(latent_channels, rev[0])]
        if with_decoder_nonlocal_attn:
            dec += [_ae_resblock(rev[0], rev[0], g, eps), _ae_attention(rev[0], g, eps), _ae_resblock(rev[0], rev[0], g, eps)]
        out_ch = rev[0]
        for i in range(n):
            in_ch, out_ch = out_ch, rev[i]
            for _ in range(rev_res[i]):
                dec.append(_ae_resblock(in_ch, out_ch, g, eps))
                in_ch = out_ch
                if rev_att[i]:
                    dec.append(_ae_attention(in_ch, g, eps))
            if i != n - 1:
                dec.append(_ae_sample(in_ch, "up"))
        dec.append(nn.GroupNorm(g, in_ch, eps=eps, affine=True))
        dec.append(_convolution(in_ch, out_channels))
        self.decoder = _Holder()
        self.decoder.blocks = nn.ModuleList(dec)
        self.quant_conv_mu = _convolution(enc_out, latent_channels, 1, 1, 0)
        self.quant_conv_log_sigma = _convolution(enc_out, latent_channels, 1, 1, 0)
        self.post_quant_conv = _convolution(latent_channels, latent_channels, 1, 1, 0)
        self._pk = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._pk = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._pk = None
        return r

    def _pack(self):
        dev = self.post_quant_conv.conv.weight.device
        if dev.type != "cuda":
            raise CcnetError("ccnet_amd.AutoencoderKL runs on the GPU only: call .to('cuda') first (no CPU fallback)")
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()

        def pconv(conv: nn.Conv2d):
            w = conv.weight.detach().float()
            co, ci, r, _ = w.shape
            return w.permute(0, 2, 3, 1).reshape(co, r * r * ci).to(BF16).contiguous(), f32(conv.bias)

        pk = {}
        for side in ("encoder", "decoder"):
            lst = []
            blocks = list(getattr(self, side).blocks)
            for i, b in enumerate(blocks):
                if isinstance(b, nn.GroupNorm):
                    lst.append(("norm", f32(b.weight), f32(b.bias)))
                elif getattr(b, "kind", None) == "res":
                    lst.append(("res", {
                        "n1": (f32(b.norm1.weight), f32(b.norm1.bias)), "n2": (f32(b.norm2.weight), f32(b.norm2.bias)),
                        "c1": pconv(b.conv1.conv), "c2": pconv(b.conv2.conv), "cout": b.cout,
                        "skip": None if isinstance(b.nin_shortcut, nn.Identity) else pconv(b.nin_shortcut.conv)}))
                elif getattr(b, "kind", None) == "attn":
                    qkv = torch.cat([b.to_q.weight, b.to_k.weight, b.to_v.weight], 0).detach()
                    lst.append(("attn", {
                        "norm": (f32(b.norm.weight), f32(b.norm.bias)), "ch": b.ch,
                        "qkv": (qkv.to(dev, BF16).contiguous(), f32(torch.cat([b.to_q.bias, b.to_k.bias, b.to_v.bias]))),
                        "proj": (b.proj_attn.weight.detach().to(dev, BF16).contiguous(), f32(b.proj_attn.bias))}))
                elif getattr(b, "kind", None) in ("down", "up"):
                    lst.append((b.kind,) + pconv(b.conv.conv))
                elif i == 0:  # first conv: tiny Cin, fp32 direct kernel
                    lst.append(("conv_first", f32(b.conv.weight), f32(b.conv.bias), b.conv.out_channels))
                else:         # last conv: tiny Cout
                    w = b.conv.weight.detach().float()
                    lst.append(("conv_last", w.permute(0, 2, 3, 1).reshape(w.shape[0], 9, w.shape[1]).to(BF16)
                                .contiguous().to(dev), f32(b.conv.bias), w.shape[0]))
            pk[side] = lst
        for nm in ("quant_conv_mu", "quant_conv_log_sigma", "post_quant_conv"):
            c = getattr(self, nm).conv
            pk[nm] = (f32(c.weight).reshape(c.out_channels, c.in_channels), f32(c.bias))
        self._pk = pk
        return pk

    def _run(self, lst, x: torch.Tensor) -> torch.Tensor:
        g, eps = self.norm_num_groups, self.norm_eps
        h = None
        for item in lst:
            kind = item[0]
            if kind == "conv_first":
                _, w, b, cout = item
                h = ops.conv_in_small(x, x.shape[0], w, b, cout, cout)
            elif kind == "res":
                p = item[1]
                t = ops.groupnorm(h, p["n1"][0], p["n1"][1], g, eps, True)
                t = ops.conv(t, p["c1"][0], p["c1"][1], p["cout"], R=3)
                t = ops.groupnorm(t, p["n2"][0], p["n2"][1], g, eps, True)
                skip = h if p["skip"] is None else ops.conv(h, p["skip"][0], p["skip"][1], p["cout"], R=1, pad=0)
                h = ops.conv(t, p["c2"][0], p["c2"][1], p["cout"], R=3, res1=skip)
            elif kind == "down":  # F.pad(x, (0,1,0,1)) + 3x3 stride-2 pad-0 conv
                h = ops.conv(h, item[1], item[2], h.C, R=3, stride=2, pad=0, out_hw=(h.H // 2, h.W // 2))
            elif kind == "up":
                h = ops.conv(h, item[1], item[2], h.C, R=3, ups=2)
            elif kind == "attn":
                p = item[1]
                C_, T = p["ch"], h.H * h.W
                if C_ % 64:
                    raise CcnetError(f"autoencoder attention over {C_} channels: the MI355X lowering needs a multiple of 64")
                t = ops.groupnorm(h, p["norm"][0], p["norm"][1], g, eps, False).t
                qkv = ops.linear(t, p["qkv"][0], p["qkv"][1], 3 * C_, C_)
                q, k, v = qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:]
                sc = 1.0 / math.sqrt(C_)
                o = ops.attention_d64(q, k, v, h.N, 1, T, T, sc) if C_ == 64 else \
                    ops.attention_generic(q, k, v, h.N, 1, T, T, C_, sc)
                h = Act(ops.linear(o, p["proj"][0], p["proj"][1], C_, C_, res1=h.t), h.N, h.H, h.W, C_)
            elif kind == "norm":  # monai-generative applies no nonlinearity after the final GroupNorm, diffusers a SiLU
                h = ops.groupnorm(h, item[1], item[2], g, eps, self.final_act)
            elif kind == "conv_last":
                return ops.conv_out_small(h, item[1], item[2], item[3])
        raise CcnetError("malformed AutoencoderKL block list")

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        pk = self._pk or self._pack()
        if not z.is_cuda:
            raise CcnetError("ccnet_amd.AutoencoderKL runs on the GPU only (no CPU fallback)")
        z = z.contiguous().float()
        zq = ops.pointwise_nchw(z, pk["post_quant_conv"][0], pk["post_quant_conv"][1], 0)
        up = 2 ** (len(self.num_channels) - 1)
        chunk = _even_chunk(z.shape[0], self._batch_chunk(z.shape[2] * up, z.shape[3] * up))
        if z.shape[0] <= chunk:
            return self._run(pk["decoder"], zq)
        return torch.cat([self._run(pk["decoder"], zq[i:i + chunk].contiguous())
                          for i in range(0, z.shape[0], chunk)], dim=0)

    def _batch_chunk(self, H: int, W: int) -> int:
        """Largest image count whose biggest activation stays below the conv kernel's 2 GiB buffer window."""
        cmax = max(self.num_channels[:2]) if len(self.num_channels) > 1 else self.num_channels[0]
        return max(1, (2 ** 31 - 2 ** 20) // (cmax * H * W * 2))

    @torch.no_grad()
    def encode(self, x: torch.Tensor):
        pk = self._pk or self._pack()
        if not x.is_cuda:
            raise CcnetError("ccnet_amd.AutoencoderKL runs on the GPU only (no CPU fallback)")
        x = x.contiguous().float()
        chunk = _even_chunk(x.shape[0], self._batch_chunk(x.shape[2], x.shape[3]))
        if x.shape[0] <= chunk:
            h = self._run(pk["encoder"], x)
        else:
            h = torch.cat([self._run(pk["encoder"], x[i:i + chunk].contiguous())
                           for i in range(0, x.shape[0], chunk)], dim=0)
        z_mu = ops.pointwise_nchw(h, pk["quant_conv_mu"][0], pk["quant_conv_mu"][1], 0)
        z_sigma = ops.pointwise_nchw(h, pk["quant_conv_log_sigma"][0], pk["quant_conv_log_sigma"][1], 1)
        return z_mu, z_sigma

    def sampling(self, z_mu: torch.Tensor, z_sigma: torch.Tensor, eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        eps = torch.randn_like(z_sigma) if eps is None else eps
        return ops.axpy_f32(z_mu, eps, z_sigma)

    def forward(self, x: torch.Tensor):
        z_mu, z_sigma = self.encode(x)
        z = self.sampling(z_mu, z_sigma)
        return self.decode(z), z_mu, z_sigma
