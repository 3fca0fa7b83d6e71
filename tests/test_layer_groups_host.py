"""The bookkeeping of `ActMaxCache`'s layer groups (which layers are grouped, when a group is launched, what a state read or an
unfinished forward does, how an in-place edit is caught) on the CPU: the three device touch-points — `ActMax.collect`,
`N.reduce_multi`, `N.actmax_update_multi` — are replaced by oracle stand-ins (checker code standing in for the kernels, in a test
only; the kernels themselves are checked in tests/test_gpu_groups.py)."""
import numpy as np
import pytest
import torch
from torch import nn

import oracle
from semanticlens_amd import _native as N
from semanticlens_amd.component_visualization import aggregators as agg
from semanticlens_amd.component_visualization.activation_caching import ActMax, ActMaxCache


class _Blocks(nn.Module):
    def __init__(self, depth=4, width=16):
        super().__init__()
        self.inp = nn.Linear(6, width)
        self.blocks = nn.ModuleList([nn.Sequential(nn.Linear(width, width), nn.Tanh()) for _ in range(depth)])
        self.odd = nn.Linear(width, 8)  # another shape: never grouped with the blocks
        self.edit_after = None
        self.stop_after = None

    def forward(self, x):
        x = self.inp(x)
        for i, b in enumerate(self.blocks):
            x = b(x)
            if self.edit_after == i:
                x.mul_(0.5)
            if self.stop_after == i:
                return x
        return self.odd(x)


@pytest.fixture
def standins(monkeypatch):
    log = []

    def oracle_of(am, C):
        if getattr(am, "_o", None) is None:
            am._o = oracle.ActMaxOracle(am.n_collect, C, oracle.MODE_ATEN)
        return am._o

    def collect(self, outs, native, id_base, site=None, k3_queue=None):
        a = oracle.agg_tokens(outs.detach().float().numpy(), "max")
        if not self.is_setup:
            self.n_latents = a.shape[1]
            self._setup_tensors()
        log.append(("single", site[1]))
        if k3_queue is not None:  # K1 now, the merge with the other layers of this forward
            k3_queue(self, torch.from_numpy(a).to(torch.bfloat16), id_base, a.shape[0])
            return
        oracle_of(self, a.shape[1]).update(a, np.arange(id_base, id_base + a.shape[0]))

    def update_one(vals, ids, cand, sample_ids, id_base, B, ties, ws):
        oracle_of(vals, cand.shape[1]).update(cand.float().numpy(), np.arange(id_base, id_base + B))
        log.append(("update_one", 1))

    def reduce_multi(kind, xs, code, pos, cand):
        for l, x in enumerate(xs):
            cand[l] = torch.from_numpy(oracle.agg_tokens(x.float().numpy(), "max")).to(torch.bfloat16)
        log.append(("reduce_multi", len(xs)))

    def update_multi(states, cands, id_bases, B):
        for (am, _), c, base in zip(states, cands, id_bases):
            oracle_of(am, c.shape[1]).update(c.float().numpy(), np.arange(base, base + B))
        log.append(("update_multi", len(states)))

    monkeypatch.setattr(ActMax, "collect", collect)
    monkeypatch.setattr(ActMax, "_device_state", lambda self, device: (self, None))
    monkeypatch.setattr(N, "reduce_multi", reduce_multi)
    monkeypatch.setattr(N, "actmax_update_multi", update_multi)
    monkeypatch.setattr(N, "actmax_update", update_one)
    monkeypatch.setattr(ActMax, "_aten_ws", lambda self, B, device: None)
    monkeypatch.setattr(N, "actmax_update_multi_supported", lambda C, k, B: True)
    return log


def _stream(model, layers, batches, log, k=4, before_batch=None, after_batch=None, edited=None, direct=False):
    cache = ActMaxCache(layers, agg.aggregate_transformer_max, n_collect=k, tie_mode="aten")
    cache._group_device_types = ("cpu",)
    mods = dict(model.named_modules())
    taps = {n: [] for n in layers}
    hs = [mods[n].register_forward_hook(lambda m, i, o, n=n: taps[n].append(o.detach().clone().numpy())) for n in layers]
    refs, seen = {}, {n: 0 for n in layers}  # ids come from a counter PER LAYER (activation_caching.py:410-413): a skipped batch does not advance it
    with torch.enable_grad(), cache.hook_context(model):  # grad mode: plain tensors with version counters
        for bi, x in enumerate(batches):
            if before_batch:
                before_batch(bi, cache)
            n_before = {n: len(taps[n]) for n in layers}
            # direct: `model.forward(x)` skips the ROOT module's hooks (the forward-boundary hook of round 6), the hooked layers'
            # own hooks still fire — the lazy bookkeeping (flush on the next forward / on a state read) must hold up alone
            (model.forward if direct else model)(x)
            for n in layers:
                if len(taps[n]) > n_before[n]:
                    a = oracle.agg_tokens(taps[n][-1] * (edited or {}).get((bi, n), 1.0), "max")
                    refs.setdefault(n, oracle.ActMaxOracle(k, a.shape[1], oracle.MODE_ATEN)).update(a, np.arange(seen[n], seen[n] + x.shape[0]))
                    seen[n] += x.shape[0]
            if after_batch:
                after_batch(bi, cache)
    for h in hs:
        h.remove()
    for n, ref in refs.items():
        got = cache.cache[n]._o
        assert np.array_equal(got.vals, ref.vals) and np.array_equal(got.ids, ref.ids), n
    return cache


def _batches(n, B=5, T=3):
    g = torch.Generator().manual_seed(0)
    return [torch.randint(0, 7, (B, T, 6), generator=g).float() / 2 for _ in range(n)]  # tie-heavy


def test_groups_are_planned_after_the_first_batch_and_launched_when_complete(standins, monkeypatch):
    monkeypatch.setenv("SEMANTICLENS_AMD_BATCH_K3", "1")  # opt-in: one merge per forward over ALL hooked layers
    torch.manual_seed(0)
    model = _Blocks().eval()
    layers = [f"blocks.{i}" for i in range(4)] + ["odd"]
    cache = _stream(model, layers, _batches(3), standins)
    assert [g["layers"] for g in cache._groups] == [[f"blocks.{i}" for i in range(4)]]
    first = [e for e in standins if e[0] == "single"][:5]
    assert [e[1] for e in first] == layers  # batch 1: every layer inside its own hook
    # batches 2 and 3: one multi reduce + one multi update for the four blocks, `odd` alone
    # ... and ONE merge for all five layers when the last one (`odd`) has fired
    assert standins.count(("reduce_multi", 4)) == 2 and standins.count(("update_multi", 5)) == 2
    assert [e for e in standins if e[0] == "single"][5:] == [("single", "odd")] * 2
    assert cache._k3_last == "odd" and not cache._k3_queue


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("batch_k3", ["0", "1"])
def test_state_read_and_unfinished_forward_flush_the_stash_layer_by_layer(standins, monkeypatch, batch_k3, direct):
    monkeypatch.setenv("SEMANTICLENS_AMD_BATCH_K3", batch_k3)
    torch.manual_seed(1)
    model = _Blocks().eval()
    layers = [f"blocks.{i}" for i in range(4)]

    def before(bi, cache):
        model.stop_after = 1 if bi == 2 else None  # batch 3 only reaches blocks.0 and blocks.1

    def after(bi, cache):
        if bi == 2:
            if direct:  # no forward boundary seen: the two outputs wait in the stash ...
                assert set(cache._groups[0]["stash"]) == {"blocks.0", "blocks.1"}
                cache.cache["blocks.1"].flush()  # ... until a state read: the whole stash is collected, layer by layer
            assert not cache._groups[0]["stash"]  # model(x): the end of the forward already collected them

    cache = _stream(model, layers, _batches(5), standins, before_batch=before, after_batch=after, direct=direct)
    model.stop_after = None
    assert standins.count(("reduce_multi", 4)) == 3  # batches 2, 4, 5
    assert cache._probe is None and [g["layers"] for g in cache._groups] == [layers]


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("batch_k3", ["0", "1"])
def test_a_forward_that_stops_early_is_finished_by_the_next_one(standins, monkeypatch, batch_k3, direct):
    monkeypatch.setenv("SEMANTICLENS_AMD_BATCH_K3", batch_k3)
    torch.manual_seed(2)
    model = _Blocks().eval()
    layers = [f"blocks.{i}" for i in range(4)]

    def before(bi, cache):
        model.stop_after = 2 if bi == 1 else None  # batch 2 never reaches blocks.3; batch 3 finds blocks.0 still stashed

    _stream(model, layers, _batches(4), standins, before_batch=before, direct=direct)
    model.stop_after = None


@pytest.mark.parametrize("batch_k3", ["0", "1"])
def test_in_place_edits(standins, monkeypatch, batch_k3):
    monkeypatch.setenv("SEMANTICLENS_AMD_BATCH_K3", batch_k3)
    torch.manual_seed(3)
    model = _Blocks().eval()
    layers = [f"blocks.{i}" for i in range(4)]
    model.edit_after = 1  # from the first batch on: blocks.1 stays in its own hook
    cache = _stream(model, layers, _batches(3), standins)
    assert [g["layers"] for g in cache._groups] == [["blocks.0", "blocks.2", "blocks.3"]]
    model.edit_after = None

    def before(bi, cache):
        model.edit_after = 2 if bi == 2 else None  # starts after the groups were planned

    # strict: the launch refuses instead of reducing modified values
    monkeypatch.setenv("SEMANTICLENS_AMD_GROUP_LAYERS", "strict")
    with pytest.raises(RuntimeError, match="modified in place after its forward hook"):
        _stream(model, layers, _batches(4), standins, before_batch=before)
    # default: ONE warning naming the samples, the edited layer leaves its group for good, every other member's batch is collected
    # intact (layer by layer), and the run goes on — the edited layer's batch 3 is what the tensor held at launch time (x 0.5)
    monkeypatch.delenv("SEMANTICLENS_AMD_GROUP_LAYERS")
    del standins[:]
    with pytest.warns(RuntimeWarning, match=r"'blocks.2' was modified in place.*samples 10\.\.14") as rec:
        cache = _stream(model, layers, _batches(5), standins, before_batch=before, edited={(2, "blocks.2"): 0.5})
    assert len([w for w in rec if "modified in place" in str(w.message)]) == 1
    assert [g["layers"] for g in cache._groups] == [["blocks.0", "blocks.1", "blocks.3"]] and "blocks.2" not in cache._group_of
    if batch_k3 == "0":
        assert standins.count(("reduce_multi", 4)) == 1 and standins.count(("reduce_multi", 3)) == 2  # batch 2; batches 4 and 5
    model.edit_after = None
    # a group left with one member is dissolved
    cache._ungroup("blocks.0")
    cache._ungroup("blocks.1")
    assert cache._group_of == {} and cache._groups[0]["layers"] == []


def test_switches(standins, monkeypatch):
    torch.manual_seed(4)
    model = _Blocks().eval()
    layers = [f"blocks.{i}" for i in range(4)]
    monkeypatch.setenv("SEMANTICLENS_AMD_GROUP_LAYERS", "0")
    cache = _stream(model, layers, _batches(3), standins)
    assert not cache._groups and not any(e[0] == "reduce_multi" for e in standins)
    monkeypatch.delenv("SEMANTICLENS_AMD_GROUP_LAYERS")
    total = ActMaxCache(layers, agg.aggregate_transformer_max, n_collect=3, tie_mode="total")
    assert not total._grouping  # the shard-invariant order keeps its own batching (one merge per 8 batches)
