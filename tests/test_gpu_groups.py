"""Layers with identical outputs collected together (round 5): one `sl_reduce_*_multi` launch over the L stashed activations and one
`sl_actmax_update_multi` launch over the L states once the last member of the group has fired — against the oracle fed from
independent taps on the same forward passes, bit for bit, and against the ungrouped path.  The reference aggregates inside every
hook (activation_caching.py:388-418); what must be preserved is its result per layer and batch, including the tie order."""
import numpy as np
import pytest
import torch
from torch import nn

import oracle
from semanticlens_amd import _native as N
from semanticlens_amd.component_visualization import aggregators as agg
from semanticlens_amd.component_visualization.activation_caching import ActMaxCache

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


class _TokenBlocks(nn.Module):
    """`depth` residual blocks over (B, T, F) tokens — the "all encoder blocks" hooking pattern of BASELINE configs[3]."""

    def __init__(self, depth=5, width=64, inplace_after=None):
        super().__init__()
        self.inp = nn.Linear(12, width)
        self.blocks = nn.ModuleList([nn.Sequential(nn.LayerNorm(width), nn.Linear(width, width), nn.GELU()) for _ in range(depth)])
        self.head = nn.Linear(width, 3)
        self.inplace_after = inplace_after  # index of a block whose output the NEXT op edits in place (when `edit` is on)
        self.edit = False

    def forward(self, x):
        x = self.inp(x)
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if self.edit and i == self.inplace_after:
                x.mul_(0.5)  # in place on the hooked output, after its hook ran
        return self.head(x.mean(1))


def _run_stream(model, layers, aggregator, kind, batches, k=7, make=None, read_mid=False, expect_launches=None):
    cache = ActMaxCache(layers, aggregator, n_collect=k, tie_mode="aten")
    mods = dict(model.named_modules())
    taps = {name: [] for name in layers}
    handles = [mods[name].register_forward_hook(lambda m, i, o, name=name: taps[name].append(o.detach().float().cpu().numpy()))
               for name in layers]
    refs = {}
    start = 0
    N.prof_enable(True)
    N.prof_reset()
    with torch.no_grad(), cache.hook_context(model):
        for bi, x in enumerate(batches):
            model(x)
            for name in layers:
                a = taps[name][-1]
                red = oracle.agg_conv(a, "max") if kind == "conv" else oracle.agg_tokens(a, "max")
                if name not in refs:
                    refs[name] = oracle.ActMaxOracle(k, red.shape[1], oracle.MODE_ATEN)
                refs[name].update(red, np.arange(start, start + a.shape[0]))
            start += x.shape[0]
            if read_mid and bi == 2:  # a state read between two forwards sees everything collected so far
                name = layers[1]
                assert np.array_equal(bits(cache.cache[name].activations), refs[name].vals)
    torch.cuda.synchronize()
    _, red_n, _ = N.prof_read(N.SL_PROF_REDUCE)
    _, mrg_n, _ = N.prof_read(N.SL_PROF_MERGE)
    N.prof_enable(False)
    for h in handles:
        h.remove()
    for name in layers:
        am = cache.cache[name]
        assert np.array_equal(bits(am.activations), refs[name].vals), name
        assert np.array_equal(am.sample_ids.numpy(), refs[name].ids), name
    if expect_launches is not None:
        assert (red_n, mrg_n) == expect_launches, (red_n, mrg_n, expect_launches)
    return cache


def _has(oracle_mod, name):
    return hasattr(oracle_mod, name)


def test_identical_token_layers_are_collected_by_one_launch_each():
    torch.manual_seed(0)
    model = _TokenBlocks(depth=5).to(DEV).eval()
    layers = [f"blocks.{i}" for i in range(5)]
    g = torch.Generator().manual_seed(1)
    sizes = (16, 16, 16, 16, 5)  # the last batch is short
    batches = [torch.randn(b, 9, 12, generator=g).to(DEV) for b in sizes]
    # batch 1 ungrouped (5 + 5 launches), then one reduce + one merge launch per batch
    cache = _run_stream(model, layers, agg.aggregate_transformer_max, "tokens", batches, expect_launches=(5 + 4, 5 + 4))
    assert len(cache._groups) == 1 and cache._groups[0]["layers"] == layers


def test_state_read_between_members_and_mid_stream():
    torch.manual_seed(1)
    model = _TokenBlocks(depth=4).to(DEV).eval()
    layers = [f"blocks.{i}" for i in range(4)]
    g = torch.Generator().manual_seed(2)
    batches = [torch.randn(8, 6, 12, generator=g).to(DEV) for _ in range(5)]
    _run_stream(model, layers, agg.aggregate_transformer_max, "tokens", batches, read_mid=True)


def test_grouping_off_gives_the_same_states(monkeypatch):
    torch.manual_seed(2)
    model = _TokenBlocks(depth=3).to(DEV).eval()
    layers = [f"blocks.{i}" for i in range(3)]
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randint(0, 5, (12, 7, 12), generator=g).float() / 2).to(DEV) for _ in range(4)]  # tie-heavy
    on = _run_stream(model, layers, agg.aggregate_transformer_max, "tokens", batches)
    monkeypatch.setenv("SEMANTICLENS_AMD_GROUP_LAYERS", "0")
    off = _run_stream(model, layers, agg.aggregate_transformer_max, "tokens", batches, expect_launches=(12, 12))
    assert not off._groups and on._groups
    for name in layers:
        assert np.array_equal(bits(on.cache[name].activations), bits(off.cache[name].activations))
        assert torch.equal(on.cache[name].sample_ids, off.cache[name].sample_ids)


def test_in_place_edit_keeps_a_layer_out_of_its_group_or_raises(monkeypatch):
    torch.manual_seed(3)
    model = _TokenBlocks(depth=4, inplace_after=1).to(DEV).eval()
    layers = [f"blocks.{i}" for i in range(4)]
    g = torch.Generator().manual_seed(4)
    batches = [torch.randn(8, 6, 12, generator=g).to(DEV) for _ in range(4)]
    # (i) the edit happens from the first batch on: blocks.1 is recognised and collected inside its own hook, the rest grouped
    model.edit = True
    cache = _run_stream(model, layers, agg.aggregate_transformer_max, "tokens", batches)
    assert [g_["layers"] for g_ in cache._groups] == [["blocks.0", "blocks.2", "blocks.3"]]
    # (ii) the edit starts AFTER the groups were planned.  strict: the launch refuses instead of reducing modified values
    model.edit = False
    monkeypatch.setenv("SEMANTICLENS_AMD_GROUP_LAYERS", "strict")
    cache = ActMaxCache(layers, agg.aggregate_transformer_max, n_collect=5, tie_mode="aten")
    with torch.no_grad(), cache.hook_context(model):
        model(batches[0])
        model(batches[1])
        model.edit = True
        with pytest.raises(RuntimeError, match="modified in place after its forward hook"):
            model(batches[2])
    model.edit = False
    # default: one warning, the edited layer leaves its group, the other members lose nothing and the run goes on
    monkeypatch.delenv("SEMANTICLENS_AMD_GROUP_LAYERS")
    cache = ActMaxCache(layers, agg.aggregate_transformer_max, n_collect=5, tie_mode="aten")
    ref = ActMaxCache(layers, agg.aggregate_transformer_max, n_collect=5, tie_mode="aten")
    ref._grouping = False  # every layer inside its own hook: sees the values before the edit
    with torch.no_grad(), cache.hook_context(model):
        model(batches[0])
        model(batches[1])
        model.edit = True
        with pytest.warns(RuntimeWarning, match="'blocks.1' was modified in place"):
            model(batches[2])
        model(batches[3])  # blocks.1 now alone in its hook (before the edit), the other three as a group
    assert [g_["layers"] for g_ in cache._groups] == [["blocks.0", "blocks.2", "blocks.3"]]
    model.edit = False
    with torch.no_grad(), ref.hook_context(model):
        for i, b in enumerate(batches):
            model.edit = i >= 2
            model(b)
    model.edit = False
    for name in ("blocks.0", "blocks.2", "blocks.3"):
        assert np.array_equal(bits(cache.cache[name].activations), bits(ref.cache[name].activations)), name
        assert torch.equal(cache.cache[name].sample_ids, ref.cache[name].sample_ids), name


def test_conv_layers_channels_last_and_nchw():
    """Identical conv outputs: channels_last maps go through the one-launch reduce, NCHW rows fall back to one reduce per tensor
    (K1's row kernel) behind the same entry point; K3 is one launch either way."""
    torch.manual_seed(4)

    class Convs(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = nn.Conv2d(3, 16, 3, padding=1)
            self.blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(16, 16, 3, padding=1), nn.ReLU()) for _ in range(3)])

        def forward(self, x):
            x = self.stem(x)
            for b in self.blocks:
                x = b(x)
            return x.mean((1, 2, 3))

    g = torch.Generator().manual_seed(5)
    layers = [f"blocks.{i}" for i in range(3)]
    for fmt, expect in ((torch.channels_last, (3 + 3, 3 + 3)), (torch.contiguous_format, (3 + 3 * 3, 3 + 3))):
        model = Convs().to(DEV).eval().to(memory_format=fmt)
        batches = [torch.randn(6, 3, 12, 12, generator=g).to(DEV).contiguous(memory_format=fmt) for _ in range(4)]
        _run_stream(model, layers, agg.aggregate_conv_max, "conv", batches, expect_launches=expect)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("code", ["max", "mean", "absmax", "absmean"])
def test_multi_reduce_equals_tensor_by_tensor(dtype, code):
    codes = {"max": N.SL_TOK_MAX, "mean": N.SL_TOK_MEAN, "absmax": N.SL_TOK_ABSMAX, "absmean": N.SL_TOK_ABSMEAN}
    g = torch.Generator().manual_seed(6)
    # 33 > one table; F = 30: fallback; (12, 8, 197, 768) and (12, 32, 197, 768): alone a tensor is below two tasks per CU (16 / 8 waves
    # split the token axis), the table of 12 is not — a sum must not change its order with the company it is reduced in (advisor, round 5)
    for L, B, T, F in ((3, 5, 17, 96), (33, 2, 9, 64), (2, 4, 50, 768), (4, 3, 8, 30), (2, 7, 197, 200), (12, 8, 197, 768), (12, 32, 197, 768)):
        xs = [torch.randn(B, T, F, generator=g).to(DEV).to(dtype) for _ in range(L)]
        xs[0][0, 1, 2] = float("nan")
        xs[-1][1, 0, 3] = float("inf")
        cand = torch.empty((L, B, F), dtype=torch.bfloat16, device=DEV)
        N.reduce_multi("tokens", xs, codes[code], 0, cand)
        for l, x in enumerate(xs):
            one = torch.empty((B, F), dtype=torch.bfloat16, device=DEV)
            N.reduce_tokens(x, codes[code], 0, one, None)
            assert np.array_equal(bits(cand[l]), bits(one)), (L, B, T, F, l)


def test_multi_update_equals_layer_by_layer():
    g = torch.Generator().manual_seed(7)
    for Cs, k, B in (([100] * 4, 20, 64), ([7, 7], 3, 5), ([16] * 33, 9, 12), ([768] * 3, 20, 256), ([512, 1024, 2048], 20, 256),
                     ([5, 300, 1, 64, 17], 7, 33), ([192, 384, 768, 1536], 100, 64)):
        L = len(Cs)
        cands = [(torch.randint(0, 9, (B, C), generator=g).float() / 4).to(torch.bfloat16).to(DEV) for C in Cs]  # tie-heavy
        multi, single = [], []
        for C in Cs:
            for store in (multi, single):
                v = torch.empty((C, k), dtype=torch.bfloat16, device=DEV)
                i = torch.empty((C, k), dtype=torch.int64, device=DEV)
                N.actmax_init(v, i)
                store.append((v, i))
        assert N.actmax_update_multi_supported(max(Cs), k, B)
        for step in range(3):
            bases = [1000 * l + step * B for l in range(L)]
            N.actmax_update_multi(multi, cands, bases, B)
            for l, (v, i) in enumerate(single):
                ws = torch.empty(N.actmax_aten_ws_bytes(Cs[l], k, B), dtype=torch.uint8, device=DEV)
                N.actmax_update(v, i, cands[l], None, bases[l], B, N.SL_TIES_ATEN, ws)
            cands = [c.roll(1, dims=0).contiguous() for c in cands]
        for (v, i), (v1, i1) in zip(multi, single):
            assert torch.equal(v.view(torch.int16), v1.view(torch.int16)) and torch.equal(i, i1)


def test_layers_of_different_shapes_share_one_merge_launch_per_forward(monkeypatch):
    """ResNet-style hooking: three layers of three shapes.  Nothing is grouped or stashed (K1 runs inside every hook), but from the
    second batch on — with SEMANTICLENS_AMD_BATCH_K3=1 — the three top-k merges of a forward run as ONE launch when the last layer has
    fired; a state read in between and the default (one merge per layer) give the same states."""
    torch.manual_seed(5)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU())
            self.b = nn.Sequential(nn.Conv2d(8, 24, 3, stride=2, padding=1), nn.ReLU())
            self.c = nn.Sequential(nn.Conv2d(24, 40, 3, stride=2, padding=1), nn.ReLU())

        def forward(self, x):
            return self.c(self.b(self.a(x))).mean((1, 2, 3))

    model = Net().to(DEV).eval()
    g = torch.Generator().manual_seed(6)
    batches = [torch.randn(b, 3, 16, 16, generator=g).to(DEV) for b in (9, 9, 9, 9, 4)]
    monkeypatch.setenv("SEMANTICLENS_AMD_BATCH_K3", "1")  # opt-in
    on = _run_stream(model, ["a", "b", "c"], agg.aggregate_conv_max, "conv", batches, read_mid=True, expect_launches=(15, 3 + 4))
    assert not on._groups and on._k3_last == "c"
    monkeypatch.delenv("SEMANTICLENS_AMD_BATCH_K3")
    off = _run_stream(model, ["a", "b", "c"], agg.aggregate_conv_max, "conv", batches, expect_launches=(15, 15))
    for name in ("a", "b", "c"):
        assert np.array_equal(bits(on.cache[name].activations), bits(off.cache[name].activations))
        assert torch.equal(on.cache[name].sample_ids, off.cache[name].sample_ids)
