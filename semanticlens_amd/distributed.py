"""Sharded concept-DB build: one process per GPU, one RCCL all-gather to merge the top-k states.

The reference is single-process; this module is the multi-GPU extension the
path admits.  Rank ``r`` of ``R`` runs the collect loop over the contiguous
sample range ``shard_range(N, r, R)`` with *global* sample ids, so every rank ends with a full
``(C, k)`` state per layer over its shard.  Top-k under the total order (value desc, id asc) is
associative and commutative, so the global state is the merge of the per-rank states:

    pack all layers' (values, ids) -> ONE all-gather (RCCL over xGMI, a few MB at most,
    latency-bound) -> K4 merge kernel on every rank -> identical state everywhere.

Who issues the collective (``COLLECTIVES`` / environment ``SL_COLLECTIVES``):

* ``"native"`` (default) — the C-ABI library itself: ``sl_actmax_allgather_merge`` (pack + ``ncclAllGather`` + K4 in one
  call) and ``sl_comm_allreduce`` on a communicator the library owns (``csrc/comm.hip``, linked against librccl).  Used
  whenever the process group's backend is ``nccl``; ``torch.distributed`` then only launches the processes and carries
  the 128-byte RCCL id from rank 0 to the others.
* ``"torch"`` — ``torch.distributed`` collectives (``all_gather_into_tensor`` / ``all_reduce``) + ``sl_actmax_merge_states``.
  The only choice under ``gloo`` (CPU tests, several ranks sharing one GPU).

No other collective is on the data path.  With ``tie_mode="total"`` the 1/2/4/8-GPU results are
bit-identical to each other.

Analysis stage (last bullet): the concept DB is small and replicated; ``text_probing_sharded`` shards
the prompt list through the text tower and the query rows through the cosine GEMM, ``eval_sharded`` shards the
component axis of clarity / polysemanticity.  Each ends in one all-gather of result rows.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

import os

from semanticlens_amd import _native as N

COLLECTIVES = os.environ.get("SL_COLLECTIVES", "native")  # "native": RCCL through the C ABI; "torch": torch.distributed
_COMMS: list = []  # [(weakref to the ProcessGroup | None for the default group, world, rank, Comm)]


def _group_of(entry):
    ref = entry[0]
    return None if ref is None else ref()


def _prune_comms():
    """Drop communicators whose process group is gone (``destroy_process_group`` + re-init, a freed subgroup whose id()
    a new one could reuse): a stale RCCL communicator with the wrong world / rank would hang or give wrong results."""
    alive = dist.is_available() and dist.is_initialized()
    keep = []
    for entry in _COMMS:
        ref, world, rank, comm = entry[:4]
        ok = alive and (ref is None or ref() is not None)
        if ok:
            g = _group_of(entry)
            try:
                ok = dist.get_world_size(g) == world and dist.get_rank(g) == rank and (g is not None or _default_pg() is entry[4])
            except Exception:
                ok = False
        if ok:
            keep.append(entry)
        else:
            try:
                comm.destroy()
            except Exception:
                pass
    _COMMS[:] = keep


def _default_pg():
    try:
        return dist.distributed_c10d._get_default_group()
    except Exception:
        return None


def native_comm(group=None, device=None):
    """The library's own RCCL communicator for ``group`` (created collectively on first use, then cached), or None when
    the collectives go through torch.distributed (``COLLECTIVES == "torch"``, or a backend other than nccl).  The cache is
    keyed on the group OBJECT (weak reference) and checked against its world size / rank on every lookup; an index-less
    ``device`` ("cuda") means the current device."""
    if COLLECTIVES != "native" or dist.get_backend(group) != "nccl":
        return None
    _prune_comms()
    for entry in _COMMS:
        if _group_of(entry) is group and (group is not None or entry[0] is None):
            return entry[3]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = N.resolve_device(device)
    box = [N.Comm.unique_id() if rank == 0 else None]
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(box, src=src, group=group, device=dev)
    comm = N.Comm(box[0], world, rank, dev)
    import weakref

    _COMMS.append((weakref.ref(group) if group is not None else None, world, rank, comm, _default_pg() if group is None else None))
    return comm


def destroy_native_comms():
    """Tear down the cached communicators (before ``dist.destroy_process_group``)."""
    while _COMMS:
        _COMMS.pop()[3].destroy()


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


def _all_reduce_host_ints(values: list[int], op, group, device) -> list[int]:
    """All-reduce a few Python ints (layer widths, cache-hit flags); device tensor under RCCL, host tensor under gloo."""
    comm = native_comm(group, device)
    if comm is not None:
        t = torch.tensor(values, dtype=torch.int64).to(comm.device)
        comm.allreduce(t, {dist.ReduceOp.MAX: "max", dist.ReduceOp.MIN: "min", dist.ReduceOp.SUM: "sum"}[op])
        return [int(v) for v in t.cpu().tolist()]
    t = torch.tensor(values, dtype=torch.int64)
    if not _host_staged(group):
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=op, group=group)
    return [int(v) for v in t.cpu().tolist()]


def merge_actmax_cache(actmax_cache, group=None, device=None):
    """Make every rank's ``ActMaxCache`` hold the global top-k (K4).  Collective call.

    A rank whose shard was empty (``N < R``, or a short last shard) never saw a batch, so its ``ActMax`` objects do
    not know their width: the layer widths are agreed first (one small MAX all-reduce) and such ranks contribute the
    initial state (values ``init_value`` = -0.0 unless configured, ids -1), which loses every comparison."""
    rank = dist.get_rank(group)
    names = list(actmax_cache.layer_names)
    widths = _all_reduce_host_ints(
        [actmax_cache.cache[n].n_latents or 0 if actmax_cache.cache[n].is_setup else 0 for n in names],
        dist.ReduceOp.MAX, group, device)
    layers = []
    for name, width in zip(names, widths):
        if width == 0:  # no rank collected anything for this layer
            continue
        am = actmax_cache.cache[name]
        if not am.is_setup:
            am.n_latents = width
            am._setup_tensors()
        elif am.n_latents != width:
            raise RuntimeError(f"layer {name!r}: this rank has {am.n_latents} components, another rank {width}")
        layers.append(name)
    if not layers:  # no rank collected anything (all widths agreed to be 0): nothing to exchange, on any rank
        return
    states = [actmax_cache.cache[name].device_state(device) for name in layers]
    comm = native_comm(group, device)
    if comm is not None:  # pack + ncclAllGather + K4 inside the library, states updated in place
        comm.actmax_allgather_merge(states)
        for name in layers:
            actmax_cache.cache[name]._dev_newer = True
        return
    world = dist.get_world_size(group)
    if states and not states[0][0].is_cuda:
        # host tensors never come out of ActMax.device_state: this branch serves tests/test_distributed_gloo.py, which
        # replaces ActMax's two device touch-points by oracle stand-ins to exercise the plumbing around them on the CPU
        gathered = all_gather_states(states, group)
        others = [r for r in range(world) if r != rank]
        for name, (vals, ids) in zip(layers, gathered):
            if others:
                actmax_cache.cache[name].merge_states(vals[others], ids[others])
        return
    mine = N.actmax_pack(states)  # the layout of pack_states; K4 reads the gathered blocks in place
    if _host_staged(group):
        host = torch.empty((world, mine.numel()), dtype=torch.uint8)
        dist.all_gather_into_tensor(host.reshape(-1), mine.cpu(), group=group)
        gathered = host.to(mine.device)
    else:
        gathered = torch.empty((world, mine.numel()), dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(gathered.reshape(-1), mine, group=group)
    N.actmax_merge_packed(states, gathered, skip_rank=rank)
    for name in layers:
        actmax_cache.cache[name]._dev_newer = True


def run_sharded(cv, batch_size: int = 64, num_workers: int = 0, group=None):
    """Sharded version of ``cv.run``: collect this rank's shard, then merge across ranks.

    ``cv`` must use ``tie_mode="total"`` (the torch.topk tie order of the reference is defined
    only for a single sequential stream).  Cache behaviour follows ``cv.run`` (activation_based.py:309-339): when
    ``cache_dir`` is set and EVERY rank finds a matching top-k cache it is returned as is; otherwise every rank starts
    from fresh states (the constructor may already have loaded a cache into ``actmax_cache`` — collecting on top of it
    would list the same samples twice, K3/K4 never de-duplicate), and rank 0 stores the merged result.
    """
    if cv.actmax_cache.tie_mode != "total":
        raise ValueError("sharded collection requires tie_mode='total'")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # a RelevanceComponentVisualizer keeps a second set of states (crp's activation mode): collected, merged and stored alike
    caches = [cv.actmax_cache] + ([cv.activation_cache] if hasattr(cv, "activation_cache") else [])
    if cv.caching:
        try:
            for cache in caches:
                cache.load(cv.storage_dir)
            hit = 1
        except FileNotFoundError:
            hit = 0
        if _all_reduce_host_ints([hit], dist.ReduceOp.MIN, group, cv.device)[0]:
            return cv.actmax_cache.cache
    for cache in caches:
        for name in cv.layer_names:  # fresh states, whatever the constructor loaded
            old = cache.cache[name]
            cache.cache[name] = type(old)(n_collect=old.n_collect, tie_mode=old.tie_mode, init_value=old.init_value)
    cv._run(batch_size=batch_size, num_workers=num_workers, sample_range=shard_range(len(cv.dataset), rank, world))
    for cache in caches:
        merge_actmax_cache(cache, group, cv.device)
    if cv.caching:
        if rank == 0:
            for cache in caches:
                cache.store(cv.storage_dir)
        dist.barrier(group=group)
    return cv.actmax_cache.cache


def gather_concept_db_sharded(embeds_local: torch.Tensor, shard_start: int, n_total: int, ids: torch.Tensor, group=None):
    """``embeds[ids]`` when rank r only holds ``embeds[shard_range(r)]``: each rank gathers the rows it
    owns (zeros elsewhere, K5 sharded form) and one all-reduce sums the disjoint pieces.  Exchanges
    ``C*k*D*4`` bytes per layer instead of the whole ``(N, D)`` table."""
    part = N.gather_rows_shard(embeds_local, ids, shard_start, n_total)
    comm = native_comm(group, part.device)
    if comm is not None:
        return comm.allreduce(part, "sum")
    if _host_staged(group) and part.is_cuda:
        host = part.cpu()
        dist.all_reduce(host, group=group)
        return host.to(part.device)
    dist.all_reduce(part, group=group)
    return part


@torch.no_grad()
def compute_concept_db_sharded(cv, fm, batch_size: int = 64, num_workers: int = 0, group=None, referenced_only: bool = False):
    """Multi-GPU ``cv._compute_concept_db(fm)``: every rank returns the same ``{layer: (C, k, D)}`` (device tensors).

    ``referenced_only``: after the merge every rank knows the global top-k ids; each rank then embeds only the
    referenced samples of its own shard instead of its whole shard."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n_total = len(cv.dataset)
    start, stop = shard_range(n_total, rank, world)
    run_sharded(cv, batch_size=batch_size, num_workers=num_workers, group=group)
    fm.to(cv.device)
    refs = {name: cv.get_max_reference(name).to(torch.int64) for name in cv.layer_names}
    if referenced_only and refs:
        flat = torch.cat([r.reshape(-1) for r in refs.values()])
        flat = torch.where(flat < 0, flat + n_total, flat)  # the -1 sentinel gathers the LAST sample
        uniq = torch.unique(flat)  # sorted global ids, identical on every rank
        lo = int(torch.searchsorted(uniq, torch.tensor(start)))
        hi = int(torch.searchsorted(uniq, torch.tensor(stop)))
        subset, n_rows, row0 = uniq[lo:hi].tolist(), uniq.numel(), lo
        refs = {name: torch.searchsorted(uniq, torch.where(r < 0, r + n_total, r)) for name, r in refs.items()}
    else:
        subset, n_rows, row0 = range(start, stop), n_total, start
    loader = torch.utils.data.DataLoader(
        torch.utils.data.Subset(cv.dataset_fm, subset), batch_size=batch_size, shuffle=False,
        collate_fn=lambda b: [i[0] if isinstance(i, (tuple, list)) else i for i in b], num_workers=num_workers,
    )
    embeds, filled = None, 0
    for items in loader:
        embeds, filled = cv.embed_batch(fm, items, embeds, filled, len(subset))
    if embeds is None:  # empty shard: still take part in the collectives
        dim = torch.zeros(1, dtype=torch.int64, device=cv.device)
    else:
        dim = torch.tensor([embeds.shape[1]], dtype=torch.int64, device=embeds.device)
    comm = native_comm(group, cv.device)
    if comm is not None:
        comm.allreduce(dim, "max")
    elif _host_staged(group) and dim.is_cuda:
        host = dim.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        dim = host
    else:
        dist.all_reduce(dim, op=dist.ReduceOp.MAX, group=group)
    if embeds is None:
        embeds = torch.empty((0, int(dim.item())), dtype=torch.float32, device=cv.device)
    return {name: gather_concept_db_sharded(embeds, row0, n_rows, ids, group) for name, ids in refs.items()}


# ------------------------------------------------------------------------------------------------
# analysis stage: shard rows, replicate the (small) concept DB, one all-gather of results
# ------------------------------------------------------------------------------------------------
def all_gather_rows(part: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Rank ``r`` holds rows ``shard_range(n_total, r, R)`` of an ``(n_total, ...)`` tensor; every rank gets all of it.
    One ``all_gather_into_tensor`` of equal blocks of ``ceil(n_total / R)`` rows (short blocks zero padded)."""
    world = dist.get_world_size(group)
    per = -(-n_total // world) if n_total else 0
    tail = tuple(part.shape[1:])
    if part.shape[0] > per:
        raise ValueError(f"local part has {part.shape[0]} rows, the shard size is {per}")
    if per == 0:
        return part.new_empty((0,) + tail)
    block = part
    if part.shape[0] < per:
        block = part.new_zeros((per,) + tail)
        block[: part.shape[0]] = part
    block = block.contiguous()
    comm = native_comm(group, block.device) if block.is_cuda else None
    if comm is not None:
        return comm.allgather(block).reshape((world * per,) + tail)[:n_total]
    if _host_staged(group) and block.is_cuda:
        host = torch.empty((world * per,) + tail, dtype=block.dtype)
        dist.all_gather_into_tensor(host, block.cpu(), group=group)
        out = host.to(block.device)
    else:
        out = torch.empty((world * per,) + tail, dtype=block.dtype, device=block.device)
        dist.all_gather_into_tensor(out, block, group=group)
    return out[:n_total]


@torch.no_grad()
def encode_text_sharded(fm, texts: list[str], batch_size: int | None = None, group=None) -> torch.Tensor:
    """``(len(texts), D)`` text embeddings on every rank; rank ``r`` runs the text tower on its slice only."""
    from semanticlens_amd.lens import _encode_texts

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(len(texts), rank, world)
    local = _encode_texts(fm, texts[lo:hi], batch_size) if hi > lo else None
    # the width agreement goes where every other collective of this module goes: the library's own communicator under
    # nccl (no second, torch-owned RCCL communicator), torch.distributed under gloo
    dim = _all_reduce_host_ints([local.shape[1] if local is not None else 0], dist.ReduceOp.MAX, group,
                                fm.device if torch.device(fm.device).type == "cuda" else None)[0]
    if local is None:
        local = torch.empty((0, dim), dtype=torch.float32, device=fm.device)
    return all_gather_rows(local.to(torch.float32), len(texts), group)


@torch.no_grad()
def probe_sharded(embeds: torch.Tensor, aggregated_concept_db, group=None, gather: bool = True):
    """``lens._probe`` (lens.py:206-214) with the query rows of ``embeds`` (replicated, ``(Q, D)``) sharded across the ranks
    against the replicated DB; the similarity rows are all-gathered so that every rank returns the single-process result
    (``gather=False``: this rank's rows ``shard_range(Q, rank, R)`` only, no collective)."""
    from semanticlens_amd.lens import _probe

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    Q, D = embeds.shape
    lo, hi = shard_range(Q, rank, world)

    def quirky(y):  # similarity_score picks its branch from the shapes (scores.py:119-128): keep those layers whole
        return y.ndim != 2 or y.shape[0] == Q or y.shape[0] == D or y.shape[1] != D

    if isinstance(aggregated_concept_db, torch.Tensor):
        if quirky(aggregated_concept_db):
            return _probe(embeds, aggregated_concept_db)
        part = _probe(embeds[lo:hi], aggregated_concept_db)
        return all_gather_rows(part, Q, group) if gather else part
    plain = {k: v for k, v in aggregated_concept_db.items() if not quirky(v)}
    local = _probe(embeds[lo:hi], plain) if plain else {}
    out = {}
    for k, v in aggregated_concept_db.items():  # keep the caller's layer order
        if k in local:
            out[k] = all_gather_rows(local[k], Q, group) if gather else local[k]
        else:
            out[k] = _probe(embeds, {k: v})[k]
    return out


@torch.no_grad()
def text_probing_sharded(fm, query, aggregated_concept_db, templates=None, batch_size=None, group=None):
    """Multi-GPU ``Lens.text_probing`` (lens.py:331-362): same result on every rank, equal to the single-process one.

    The (templated) prompt list is sharded through the text tower and all-gathered, the template mean (which mixes
    prompts of different queries, SURVEY.md finding 4) runs on the full list, then query rows are sharded through the
    cosine GEMM against the replicated DB and the similarity rows are all-gathered (``probe_sharded``)."""
    from semanticlens_amd.lens import _embed_text_probes

    queries = query if isinstance(query, list) else [query]
    embeds = _embed_text_probes(fm, queries, templates, batch_size,
                                encode=lambda texts, bs: encode_text_sharded(fm, texts, bs, group))
    return probe_sharded(embeds, aggregated_concept_db, group)


@torch.no_grad()
def eval_sharded(score_fn, concept_db, group=None):
    """Per-component scores (``clarity_score`` / ``polysemanticity_score``: every component is independent) with the
    component axis sharded across ranks; ``concept_db``: ``(C, n, D)`` tensor or dict of them, replicated.

    A dict of ``(C_l, n, D)`` layers costs ONE exchange: rank r scores rows ``shard_range(C_l, r, R)`` of every layer (clarity: all
    layers in one K7 launch, like ``Lens.eval_clarity``), the per-layer blocks (padded to ``ceil(C_l / R)``) travel in one
    all-gather, and every rank cuts the layers back out."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def one(V):
        C = V.shape[0]
        lo, hi = shard_range(C, rank, world)
        if hi > lo:
            part = score_fn(V[lo:hi])
        else:
            part = None
        ref = score_fn(V[:1]) if part is None and C > 0 else part  # dtype/device of the result for an empty shard
        if part is None:
            part = ref[:0] if ref is not None else torch.empty(0)
        return all_gather_rows(part, C, group)

    if isinstance(concept_db, torch.Tensor):
        return one(concept_db)
    keys = list(concept_db)
    if len(keys) < 2 or not all(isinstance(concept_db[k], torch.Tensor) and concept_db[k].ndim == 3 and concept_db[k].shape[0] > 0
                                and concept_db[k].device == concept_db[keys[0]].device for k in keys):
        return {k: one(v) for k, v in concept_db.items()}
    from semanticlens_amd.scores import clarity_score

    cuts = {k: shard_range(concept_db[k].shape[0], rank, world) for k in keys}
    mine = [k for k in keys if cuts[k][1] > cuts[k][0]]
    parts = {}
    if score_fn is clarity_score and len(mine) > 1:
        outs = N.clarity_multi([concept_db[k][cuts[k][0]:cuts[k][1]] for k in mine])
        if outs is not None:
            parts = dict(zip(mine, outs))
    for k in mine:
        if k not in parts:
            parts[k] = score_fn(concept_db[k][cuts[k][0]:cuts[k][1]])
    probe = next(iter(parts.values())) if parts else score_fn(concept_db[keys[0]][:1])
    pers = [-(-concept_db[k].shape[0] // world) for k in keys]
    block = torch.zeros(sum(pers), dtype=probe.dtype, device=concept_db[keys[0]].device)
    off = 0
    for k, per in zip(keys, pers):
        if k in parts:
            block[off:off + parts[k].shape[0]] = parts[k].to(block.device)
        off += per
    full = all_gather_rows(block.reshape(1, -1), world, group)  # (R, sum of per-layer blocks)
    out, off = {}, 0
    for k, per in zip(keys, pers):
        out[k] = full[:, off:off + per].reshape(-1)[: concept_db[k].shape[0]].contiguous()
        off += per
    return out
