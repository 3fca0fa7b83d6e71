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

from semanticlens_amd import _native as N


def shard_range(n_samples: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous ``[start, stop)`` of rank ``rank``: blocks of ``ceil(N / R)`` samples."""
    per = -(-n_samples // world_size)
    start = min(n_samples, rank * per)
    return start, min(n_samples, start + per)


def pack_states(states: list[tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """[(vals (C,k) bf16, ids (C,k) int64), ...] -> one flat uint8 tensor (ids first: 8-byte aligned)."""
    parts = [ids.contiguous().view(torch.uint8).reshape(-1) for _, ids in states]
    parts += [vals.contiguous().view(torch.uint8).reshape(-1) for vals, _ in states]
    if not parts:
        return torch.empty(0, dtype=torch.uint8)
    n = sum(p.numel() for p in parts)
    pad = (-n) % 16  # keep every rank's slice of the gathered buffer 16-byte aligned
    if pad:
        parts.append(torch.zeros(pad, dtype=torch.uint8, device=parts[0].device))
    return torch.cat(parts)


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


def _host_staged(group) -> bool:
    """gloo cannot move device tensors in every collective: stage through the host (tests / debugging only;
    the production backend is "nccl" = RCCL, which works on device memory directly)."""
    return dist.get_backend(group) == "gloo"


def all_gather_states(states, group=None):
    """All-gather every rank's packed states.  Returns ``[per-layer (vals (R,C,k), ids (R,C,k))]``."""
    world = dist.get_world_size(group)
    shapes = [tuple(v.shape) for v, _ in states]
    mine = pack_states(states)
    if _host_staged(group) and mine.is_cuda:
        host = torch.empty((world, mine.numel()), dtype=torch.uint8)
        dist.all_gather_into_tensor(host.reshape(-1), mine.cpu(), group=group)
        gathered = host.to(mine.device)
    else:
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


def gather_concept_db_sharded(embeds_local: torch.Tensor, shard_start: int, n_total: int, ids: torch.Tensor, group=None):
    """``embeds[ids]`` when rank r only holds ``embeds[shard_range(r)]``: each rank gathers the rows it
    owns (zeros elsewhere, K5 sharded form) and one all-reduce sums the disjoint pieces.  Exchanges
    ``C*k*D*4`` bytes per layer instead of the whole ``(N, D)`` table."""
    part = N.gather_rows_shard(embeds_local, ids, shard_start, n_total)
    if _host_staged(group) and part.is_cuda:
        host = part.cpu()
        dist.all_reduce(host, group=group)
        return host.to(part.device)
    dist.all_reduce(part, group=group)
    return part


@torch.no_grad()
def compute_concept_db_sharded(cv, fm, batch_size: int = 64, num_workers: int = 0, group=None):
    """Multi-GPU ``cv._compute_concept_db(fm)``: every rank returns the same ``{layer: (C, k, D)}`` (device tensors)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n_total = len(cv.dataset)
    start, stop = shard_range(n_total, rank, world)
    run_sharded(cv, batch_size=batch_size, num_workers=num_workers, group=group)
    fm.to(cv.device)
    loader = torch.utils.data.DataLoader(
        torch.utils.data.Subset(cv.dataset_fm, range(start, stop)), batch_size=batch_size, shuffle=False,
        collate_fn=lambda b: [i[0] if isinstance(i, (tuple, list)) else i for i in b], num_workers=num_workers,
    )
    embeds, filled = None, 0
    for items in loader:
        embeds, filled = cv.embed_batch(fm, items, embeds, filled, stop - start)
    if embeds is None:  # empty shard: still take part in the collectives
        dim = torch.zeros(1, dtype=torch.int64, device=cv.device)
    else:
        dim = torch.tensor([embeds.shape[1]], dtype=torch.int64, device=embeds.device)
    dist.all_reduce(dim, op=dist.ReduceOp.MAX, group=group)
    if embeds is None:
        embeds = torch.empty((0, int(dim.item())), dtype=torch.float32, device=cv.device)
    return {
        name: gather_concept_db_sharded(embeds, start, n_total, cv.get_max_reference(name), group)
        for name in cv.layer_names
    }
