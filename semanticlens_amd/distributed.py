"""Sharded concept-DB build: one process per GPU, one RCCL all-gather to merge the top-k states.

The reference is single-process; this module is the multi-GPU extension the
path admits.  Rank ``r`` of ``R`` runs the collect loop over the contiguous
sample range ``shard_range(N, r, R)`` with *global* sample ids, so every rank ends with a full
``(C, k)`` state per layer over its shard.  Top-k under the total order (value desc, id asc) is
associative and commutative, so the global state is the merge of the per-rank states:

    pack all layers' (values, ids) -> ONE ``all_gather_into_tensor`` (RCCL over xGMI, a few MB
    at most, latency-bound) -> K4 merge kernel on every rank -> identical state everywhere.

No other collective is on the data path.  With ``tie_mode="total"`` the 1/2/4/8-GPU results are
bit-identical to each other.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_samples: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous ``[start, stop)`` of rank ``rank``: blocks of ``ceil(N / R)`` samples."""
    per = -(-n_samples // world_size)
    start = min(n_samples, rank * per)
    return start, min(n_samples, start + per)


def pack_states(states: list[tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """[(vals (C,k) bf16, ids (C,k) int64), ...] -> one flat uint8 tensor (ids first: 8-byte aligned)."""
    parts = [ids.contiguous().view(torch.uint8).reshape(-1) for _, ids in states]
    parts += [vals.contiguous().view(torch.uint8).reshape(-1) for vals, _ in states]
    return torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8)


def unpack_states(buf: torch.Tensor, shapes: list[tuple[int, int]]):
    """Inverse of :func:`pack_states` for one rank's buffer."""
    out_ids, out_vals = [], []
    off = 0
    for C, k in shapes:
        n = C * k * 8
        out_ids.append(buf[off : off + n].view(torch.int64).reshape(C, k))
        off += n
    for C, k in shapes:
        n = C * k * 2
        out_vals.append(buf[off : off + n].view(torch.bfloat16).reshape(C, k))
        off += n
    return list(zip(out_vals, out_ids))


def all_gather_states(states, group=None):
    """All-gather every rank's packed states.  Returns ``[per-layer (vals (R,C,k), ids (R,C,k))]``."""
    world = dist.get_world_size(group)
    shapes = [tuple(v.shape) for v, _ in states]
    mine = pack_states(states)
    gathered = torch.empty((world, mine.numel()), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered.reshape(-1), mine, group=group)
    per_rank = [unpack_states(gathered[r], shapes) for r in range(world)]
    out = []
    for li in range(len(shapes)):
        out.append(
            (torch.stack([per_rank[r][li][0] for r in range(world)]), torch.stack([per_rank[r][li][1] for r in range(world)]))
        )
    return out


def merge_actmax_cache(actmax_cache, group=None):
    """Make every rank's ``ActMaxCache`` hold the global top-k (K4).  Collective call."""
    rank = dist.get_rank(group)
    layers = [name for name in actmax_cache.layer_names if actmax_cache.cache[name].is_setup]
    states = [actmax_cache.cache[name].device_state() for name in layers]
    gathered = all_gather_states(states, group)
    world = dist.get_world_size(group)
    others = [r for r in range(world) if r != rank]
    for name, (vals, ids) in zip(layers, gathered):
        if others:
            actmax_cache.cache[name].merge_states(vals[others], ids[others])


def run_sharded(cv, batch_size: int = 64, num_workers: int = 0, group=None):
    """Sharded version of ``cv.run``: collect this rank's shard, then merge across ranks.

    ``cv`` must use ``tie_mode="total"`` (the torch.topk tie order of the reference is defined
    only for a single sequential stream).
    """
    if cv.actmax_cache.tie_mode != "total":
        raise ValueError("sharded collection requires tie_mode='total'")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    cv._run(batch_size=batch_size, num_workers=num_workers, sample_range=shard_range(len(cv.dataset), rank, world))
    merge_actmax_cache(cv.actmax_cache, group)
    return cv.actmax_cache.cache
