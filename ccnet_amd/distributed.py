"""Data-parallel sampling: seed-indexed shards, one process per GPU, one all-gather of the outputs.

The reference samples on one GPU at batch 1 (run_controlnet_inference.py:93,212) and has no collective on this
path.  Images are independent units (own noise / cond / text / acquisition time, no
batch-coupled op), so the path shards with NO data-path collective; the only exchange is one all-gather of the
generated latents and uint8 slices for downstream FRD/FID evaluation (RCCL over xGMI when the backend is
"nccl"; "gloo" on CPU for tests).  Noise is addressed by sample index, so results do not depend on the world
size or on how indices are packed into batches.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import torch


def shard_indices(n_total: int, world_size: int, rank: int) -> range:
    """Contiguous block [r*ceil(N/W), min(N, (r+1)*ceil(N/W)))  (SURVEY.md section 8(e))."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    per = (n_total + world_size - 1) // world_size
    return range(min(n_total, rank * per), min(n_total, (rank + 1) * per))


def batches(indices: Sequence[int], batch_size: int) -> List[List[int]]:
    idx = list(indices)
    return [idx[i:i + batch_size] for i in range(0, len(idx), batch_size)]


def gather_outputs(local: Dict[str, torch.Tensor], n_total: int, world_size: int, rank: int,
                   dist=None) -> Dict[str, torch.Tensor]:
    """All-gather per-rank outputs ([n_local, ...] each, in index order) into [n_total, ...] on every rank.
    Shards are padded to ceil(N/W) rows so that one `all_gather_into_tensor` per tensor suffices."""
    if world_size == 1:
        return {k: v[:n_total] for k, v in local.items()}
    per = (n_total + world_size - 1) // world_size
    out = {}
    for k, v in local.items():
        pad = torch.zeros((per,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: v.shape[0]] = v
        full = torch.empty((world_size * per,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        dist.all_gather_into_tensor(full, pad)
        out[k] = full[:n_total]
    return out


def sample_sharded(run_batch: Callable[[List[int]], Dict[str, torch.Tensor]], n_total: int, batch_size: int,
                   world_size: int, rank: int, dist=None) -> Tuple[Dict[str, torch.Tensor], range]:
    """Run `run_batch(indices) -> {"latent": [b,...], "image_u8": [b,...]}` over this rank's shard and gather."""
    mine = shard_indices(n_total, world_size, rank)
    parts: Dict[str, List[torch.Tensor]] = {}
    for b in batches(mine, batch_size):
        r = run_batch(b)
        for k, v in r.items():
            parts.setdefault(k, []).append(v)
    local = {k: torch.cat(v, dim=0) for k, v in parts.items()}
    return gather_outputs(local, n_total, world_size, rank, dist), mine
