"""Data-parallel sampling: seed-indexed shards, one process per GPU, one all-gather of the outputs.

The reference samples on one GPU at batch 1 (run_controlnet_inference.py:93,212) and has no collective on this
path.  Images are independent units (own noise / cond / text / acquisition time, no
batch-coupled op), so the path shards with NO data-path collective; the only exchange is one all-gather of the
generated latents and uint8 slices for downstream FRD/FID evaluation (RCCL over xGMI when the backend is
"nccl"; "gloo" on CPU for tests).  Noise is addressed by sample index, so a sample's inputs do not depend on the world
size or on how indices are packed into batches; its output is bit-identical at equal batch size (any rank, any position in
the batch) and agrees to bf16 rounding across batch sizes (tile plans depend on the row count).
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


def _agree_spec(local: Dict[str, torch.Tensor], world_size: int, dist, spec=None):
    """{key: (trailing shape, dtype)} that EVERY rank uses for the collectives below.  Ceil-division sharding leaves
    trailing ranks without a single sample whenever N is not close to a multiple of W (N = 9, W = 8: ranks 5-7); such a
    rank has no tensors to read the layout from, yet it must enter every all-gather or the others block forever.  Rank 0
    always owns a non-empty shard when N >= 1, so its layout is broadcast (one small object collective per call)."""
    if spec is None:
        box = [{k: (tuple(v.shape[1:]), v.dtype) for k, v in local.items()}]
        if world_size > 1:
            dist.broadcast_object_list(box, src=0)
        spec = box[0]
    return {k: spec[k] for k in sorted(spec)}


def gather_outputs(local: Dict[str, torch.Tensor], n_total: int, world_size: int, rank: int,
                   dist=None, spec=None, device=None) -> Dict[str, torch.Tensor]:
    """All-gather per-rank outputs ([n_local, ...] each, in index order) into [n_total, ...] on every rank.
    Shards are padded to ceil(N/W) rows so that one `all_gather_into_tensor` per tensor suffices; a rank with an empty
    shard contributes an all-padding block (see _agree_spec).  `spec`: optional {key: (trailing shape, dtype)} known to the
    caller, which saves the layout broadcast; `device`: where an empty rank allocates its padding (default: the device of
    its local tensors, else the current GPU when the backend is RCCL, else the CPU)."""
    if world_size == 1:
        return {k: v[:n_total] for k, v in local.items()}
    spec = _agree_spec(local, world_size, dist, spec)
    per = (n_total + world_size - 1) // world_size
    if device is None:
        if local:
            device = next(iter(local.values())).device
        else:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    out = {}
    for k, (tail, dtype) in spec.items():  # same keys in the same (sorted) order on every rank
        v = local.get(k)
        if v is not None and (tuple(v.shape[1:]) != tuple(tail) or v.dtype != dtype):
            raise ValueError(f"rank {rank}: output {k!r} is {tuple(v.shape[1:])} {v.dtype}, rank 0 has {tuple(tail)} {dtype}")
        pad = torch.zeros((per,) + tuple(tail), dtype=dtype, device=device)
        if v is not None:
            pad[: v.shape[0]] = v
        full = torch.empty((world_size * per,) + tuple(tail), dtype=dtype, device=device)
        dist.all_gather_into_tensor(full, pad)
        out[k] = full[:n_total]
    return out


def sample_sharded(run_batch: Callable[[List[int]], Dict[str, torch.Tensor]], n_total: int, batch_size: int,
                   world_size: int, rank: int, dist=None, spec=None, device=None) -> Tuple[Dict[str, torch.Tensor], range]:
    """Run `run_batch(indices) -> {"latent": [b,...], "image_u8": [b,...]}` over this rank's shard and gather.  Every rank
    returns the same full, index-ordered dict, also a rank whose shard is empty."""
    mine = shard_indices(n_total, world_size, rank)
    parts: Dict[str, List[torch.Tensor]] = {}
    for b in batches(mine, batch_size):
        r = run_batch(b)
        for k, v in r.items():
            parts.setdefault(k, []).append(v)
    local = {k: torch.cat(v, dim=0) for k, v in parts.items()}
    return gather_outputs(local, n_total, world_size, rank, dist, spec, device), mine
