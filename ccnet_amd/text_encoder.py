"""CLIP text tower on the MI355X kernels: the `transformers.CLIPTextModel` surface the reference calls.

Reference call sites: `CLIPTextModel.from_pretrained(args.source_model, subfolder="text_encoder")`
(run_controlnet_inference.py:126, train_controlnet.py:318), `.to(device)`, `.eval()`, `text_encoder(input_ids)[0]`
(run_controlnet_inference.py:149-150,205; training_functions.py:540,927,1087).  Parameters live in an nn.Module tree
with transformers' key names, so checkpoints of either transformers generation load (`text_model.` prefix optional).

Lowering (per encoder layer, rows = B*S tokens, bf16 activations, fp32 accumulation):
    LayerNorm -> fused q|k|v linear (ccnet_conv_gemm_bf16) -> causal flash attention (ccnet_attention_d64_causal_bf16,
    scale head_dim^-0.5) -> out_proj + residual (conv epilogue) -> LayerNorm -> fc1 -> GELU (ccnet_gelu_bf16) ->
    fc2 + residual.
Embedding lookup is ccnet_embed_tokens_bf16, the final LayerNorm ccnet_layernorm_bf16.  There is no CPU path.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .diffusers_dialect import FrozenConfig
from .lib import CcnetError

_DEFAULTS = dict(vocab_size=49408, hidden_size=512, intermediate_size=2048, projection_dim=512, num_hidden_layers=12,
                 num_attention_heads=8, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                 attention_dropout=0.0, initializer_range=0.02, initializer_factor=1.0, pad_token_id=1, bos_token_id=49406,
                 eos_token_id=49407)


@dataclass
class TextModelOutput:
    """`BaseModelOutputWithPooling` stand-in: attribute access and `out[0]` / `out[1]` indexing."""
    last_hidden_state: torch.Tensor
    pooler_output: torch.Tensor

    def __getitem__(self, i):
        return (self.last_hidden_state, self.pooler_output)[i]

    def to_tuple(self):
        return (self.last_hidden_state, self.pooler_output)


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(c, c) for _ in range(4))


class _MLP(nn.Module):
    def __init__(self, c, inner):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(c, inner), nn.Linear(inner, c)


class _Layer(nn.Module):
    def __init__(self, c, inner, eps):
        super().__init__()
        self.self_attn = _Attn(c)
        self.layer_norm1 = nn.LayerNorm(c, eps=eps)
        self.mlp = _MLP(c, inner)
        self.layer_norm2 = nn.LayerNorm(c, eps=eps)


class _Embeddings(nn.Module):
    def __init__(self, vocab, npos, c):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, c)
        self.position_embedding = nn.Embedding(npos, c)


class _Encoder(nn.Module):
    def __init__(self, n, c, inner, eps):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(c, inner, eps) for _ in range(n)])


class _TextTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg.vocab_size, cfg.max_position_embeddings, cfg.hidden_size)
        self.encoder = _Encoder(cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size, cfg.layer_norm_eps)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPTextModel(nn.Module):
    """Drop-in for `transformers.CLIPTextModel` on the path the reference uses (input_ids -> last_hidden_state)."""

    def __init__(self, config=None, **kwargs):
        super().__init__()
        c = dict(_DEFAULTS)
        c.update(dict(config) if config is not None else {})
        c.update(kwargs)
        self.config = cfg = FrozenConfig(c)
        if cfg.hidden_size % cfg.num_attention_heads or cfg.hidden_size // cfg.num_attention_heads != 64:
            raise NotImplementedError("text tower: head dim must be 64 (CLIP ViT-B/L/H text towers all are)")
        if cfg.hidden_size % 64 or cfg.intermediate_size % 64:
            raise NotImplementedError("text tower: hidden and intermediate sizes must be multiples of 64")
        if cfg.hidden_act not in ("gelu", "quick_gelu"):
            raise NotImplementedError(f"text tower: hidden_act {cfg.hidden_act!r}")
        self.text_model = _TextTransformer(cfg)
        self._packed = None

    # ---- checkpoints: keys with or without the `text_model.` prefix (transformers 4.x / 5.x), position_ids ignored
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {}
        for k, v in dict(state_dict).items():
            if k.endswith("position_ids"):
                continue
            sd[k if k.startswith("text_model.") else "text_model." + k] = v
        self._packed = None
        return super().load_state_dict(sd, strict=strict, **kw)

    @classmethod
    def from_config(cls, config, **kw):
        return cls({k: v for k, v in dict(config).items() if not k.startswith("_")}, **kw)

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, **kw):
        """LOCAL directories only: `<path>[/<subfolder>]/config.json` + `model.safetensors` (or `pytorch_model.bin`,
        loaded with weights_only=True).  Hub ids cannot be resolved here (no network) and raise."""
        d = os.path.join(str(path), subfolder) if subfolder else str(path)
        if not os.path.isdir(d):
            raise OSError(f"{d} is not a local directory: hub model ids cannot be fetched (no network access); "
                          "download the folder and pass its path")
        with open(os.path.join(d, "config.json")) as fh:
            raw = json.load(fh)
        raw = raw.get("text_config", raw) if "hidden_size" not in raw else raw
        model = cls({k: v for k, v in raw.items() if k in _DEFAULTS}, **kw)
        st = os.path.join(d, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd)
        return model.eval()

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    @property
    def dtype(self):
        return torch.float32

    def get_input_embeddings(self):
        return self.text_model.embeddings.token_embedding

    # ---- lowering
    def _pack(self):
        """bf16 GEMM weights (q|k|v fused), fp32 biases / norm parameters / embedding tables; rebuilt when a parameter
        changed (torch's in-place version counters) or moved."""
        tm = self.text_model
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is not None and self._packed["key"] == key:
            return self._packed
        bf, f32 = torch.bfloat16, torch.float32
        layers = []
        with torch.no_grad():
            for l in tm.encoder.layers:
                a = l.self_attn
                layers.append(dict(
                    ln1=(l.layer_norm1.weight.to(f32).contiguous(), l.layer_norm1.bias.to(f32).contiguous()),
                    ln2=(l.layer_norm2.weight.to(f32).contiguous(), l.layer_norm2.bias.to(f32).contiguous()),
                    wqkv=torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight]).to(bf).contiguous(),
                    bqkv=torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias]).to(f32).contiguous(),
                    wo=a.out_proj.weight.to(bf).contiguous(), bo=a.out_proj.bias.to(f32).contiguous(),
                    w1=l.mlp.fc1.weight.to(bf).contiguous(), b1=l.mlp.fc1.bias.to(f32).contiguous(),
                    w2=l.mlp.fc2.weight.to(bf).contiguous(), b2=l.mlp.fc2.bias.to(f32).contiguous()))
            self._packed = dict(
                key=key, layers=layers,
                tok=tm.embeddings.token_embedding.weight.to(f32).contiguous(),
                pos=tm.embeddings.position_embedding.weight.to(f32).contiguous(),
                lnf=(tm.final_layer_norm.weight.to(f32).contiguous(), tm.final_layer_norm.bias.to(f32).contiguous()))
        return self._packed

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask=None, position_ids=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, **unused):
        cfg = self.config
        if attention_mask is not None or position_ids is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("text tower: only text_encoder(input_ids) is lowered (the reference's call)")
        if not input_ids.is_cuda or not self.device.type == "cuda":
            raise CcnetError("text tower runs on the GPU only: move the model and input_ids to cuda")
        ids = input_ids.reshape(-1, input_ids.shape[-1]).to(torch.int64).contiguous()
        B, S = ids.shape
        if S > cfg.max_position_embeddings:
            raise ValueError(f"sequence length {S} > max_position_embeddings {cfg.max_position_embeddings}")
        if int(ids.min()) < 0 or int(ids.max()) >= cfg.vocab_size:
            raise IndexError("input_ids outside the vocabulary")
        pk = self._pack()
        Cc, H = cfg.hidden_size, cfg.num_attention_heads
        quick = cfg.hidden_act == "quick_gelu"
        x = ops.embed_tokens(ids.view(-1), S, pk["tok"], pk["pos"])
        for l in pk["layers"]:
            h = ops.layernorm(x, Cc, *l["ln1"], eps=cfg.layer_norm_eps)
            qkv = ops.linear(h, l["wqkv"], l["bqkv"], 3 * Cc, Cc)
            a = ops.attention_d64_causal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], B, H, S, 64 ** -0.5)
            x = ops.linear(a, l["wo"], l["bo"], Cc, Cc, res1=x)
            h = ops.layernorm(x, Cc, *l["ln2"], eps=cfg.layer_norm_eps)
            h = ops.gelu(ops.linear(h, l["w1"], l["b1"], cfg.intermediate_size, Cc), cfg.intermediate_size, quick)
            x = ops.linear(h, l["w2"], l["b2"], Cc, cfg.intermediate_size, res1=x)
        x = ops.layernorm(x, Cc, *pk["lnf"], eps=cfg.layer_norm_eps).float().view(B, S, Cc)
        if cfg.eos_token_id == 2:
            eos = ids.argmax(dim=-1)
        else:
            eos = (ids == cfg.eos_token_id).int().argmax(dim=-1)
        out = TextModelOutput(x, x[torch.arange(B, device=x.device), eos])
        return out if return_dict is None or return_dict else out.to_tuple()
