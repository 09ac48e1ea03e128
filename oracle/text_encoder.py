"""fp32 restatement of the CLIP text tower the reference runs its prompts through.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned: tests/test_cpu.py checks it against `transformers`'
own CLIPTextModel (the third-party module the reference imports, importable in this image) on seeded random weights,
for both activations and both pooling rules.

Reference call sites this follows:
  * CLIPTextModel.from_pretrained(source_model, subfolder="text_encoder")   -- run_controlnet_inference.py:126
  * empty_prompt_embeds = text_encoder(input_ids)[0]                          -- run_controlnet_inference.py:149-150
  * prompt_embeds = text_encoder(reports.squeeze(1))[0]                       -- run_controlnet_inference.py:205,
    training_functions.py:540,927,1087
Upstream algorithm: huggingface/transformers models/clip/modeling_clip.py (CLIPTextTransformer): token + position
embeddings, pre-LN encoder layers with causal self-attention (q scaled by head_dim^-0.5) and a 2-layer MLP,
final_layer_norm; pooled output = hidden state at the EOS token.  State-dict keys are transformers' own
(`[text_model.]embeddings.token_embedding.weight`, `encoder.layers.N.self_attn.{q,k,v,out}_proj`, `layer_norm1/2`,
`mlp.fc1/fc2`, `final_layer_norm`).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


def strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers < 5 prefixes every key with `text_model.`; drop it and the non-parameter position_ids buffer."""
    out = {}
    for k, v in sd.items():
        k = k[len("text_model."):] if k.startswith("text_model.") else k
        if k.endswith("position_ids"):
            continue
        out[k] = v
    return out


def clip_text_forward(sd: Dict[str, torch.Tensor], input_ids: torch.Tensor, num_heads: int, hidden_act: str = "gelu",
                      eps: float = 1e-5, eos_token_id: int = 2) -> Tuple[torch.Tensor, torch.Tensor]:
    """(last_hidden_state [B,S,C], pooler_output [B,C]) in fp32."""
    sd = {k: v.float() for k, v in strip_prefix(sd).items()}
    B, S = input_ids.shape
    x = sd["embeddings.token_embedding.weight"][input_ids] + sd["embeddings.position_embedding.weight"][:S]
    Cc = x.shape[-1]
    d = Cc // num_heads
    mask = torch.full((S, S), float("-inf")).triu(1)
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (Cc,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * d ** -0.5
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(B, S, num_heads, d).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, S, Cc)
        x = x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (Cc,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
        h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        if hidden_act == "gelu":
            h = 0.5 * h * (1.0 + torch.erf(h / math.sqrt(2.0)))
        elif hidden_act == "quick_gelu":
            h = h * torch.sigmoid(1.702 * h)
        else:
            raise NotImplementedError(hidden_act)
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = F.layer_norm(x, (Cc,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], eps)
    return x, x[torch.arange(B), eos_positions(input_ids, eos_token_id)]


def eos_positions(input_ids: torch.Tensor, eos_token_id: int) -> torch.Tensor:
    """modeling_clip.py: configs with the legacy eos_token_id == 2 take the highest token id of each row (the EOS token
    is the largest id of the CLIP vocabulary); newer configs take the first position that holds eos_token_id."""
    if eos_token_id == 2:
        return input_ids.argmax(dim=-1)
    return (input_ids == eos_token_id).int().argmax(dim=-1)
