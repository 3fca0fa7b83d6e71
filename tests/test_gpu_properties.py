"""Property-based parity (hypothesis) for the streaming top-k (K3) and the cross-rank merge (K4): random streams with
heavy ties, negative and special values, arbitrary batch splits (empty batches included) and k around the stream length,
against the oracle in both tie modes; total mode additionally against itself under re-batching and sharding — the
size-independent properties SURVEY.md §8c names (batch invariance, mergeability, sortedness)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import oracle
from semanticlens_amd.component_visualization.activation_caching import ActMax

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COMMON = dict(deadline=None, max_examples=40, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


@st.composite
def streams(draw):
    n = draw(st.integers(1, 400))
    c = draw(st.integers(1, 70))
    k = draw(st.integers(1, 40))
    seed = draw(st.integers(0, 2**31 - 1))
    kind = draw(st.sampled_from(["ties", "relu", "signed", "special", "constant"]))
    rng = np.random.RandomState(seed)
    if kind == "ties":
        a = rng.randint(0, 6, size=(n, c)).astype(np.float32) / 4
    elif kind == "relu":
        a = np.maximum(rng.randn(n, c), 0).astype(np.float32)
    elif kind == "signed":
        a = rng.randn(n, c).astype(np.float32)
    elif kind == "constant":
        a = np.full((n, c), float(rng.randint(-1, 3)), dtype=np.float32)
    else:
        a = rng.randn(n, c).astype(np.float32)
        for _ in range(min(12, n * c)):
            a[rng.randint(n), rng.randint(c)] = rng.choice([np.inf, -np.inf, 0.0, -0.0, 1e-40, 3.4e38])
    cuts = sorted(draw(st.lists(st.integers(0, n), max_size=8)))
    bounds = [0] + cuts + [n]
    return a, k, bounds


@settings(**COMMON)
@given(streams(), st.sampled_from(["aten", "total"]))
def test_streaming_topk_equals_oracle_for_any_batching(stream, mode):
    acts, k, bounds = stream
    n, c = acts.shape
    am = ActMax(n_collect=k, n_latents=c, tie_mode=mode)
    ref = oracle.ActMaxOracle(k, c, oracle.MODE_ATEN if mode == "aten" else oracle.MODE_TOTAL)
    x = torch.from_numpy(acts).to(DEV)
    for s, e in zip(bounds[:-1], bounds[1:]):
        if e == s:
            continue  # the reference never sees an empty batch (a DataLoader does not yield one)
        am.update(x[s:e], torch.arange(s, e))
        ref.update(acts[s:e], np.arange(s, e))
    assert np.array_equal(bits(am.activations), ref.vals)
    assert np.array_equal(am.sample_ids.numpy(), ref.ids)
    # sortedness: values non-increasing along k (a 0.0 activation ties with the -0.0 sentinel of an unfilled slot, so
    # filled and unfilled slots may interleave at the value 0 exactly as in the reference)
    v = oracle.bf16_to_f32(bits(am.activations))
    if k > 1:
        assert np.all(v[:, :-1] >= v[:, 1:])


@settings(**COMMON)
@given(streams(), st.integers(2, 5))
def test_total_mode_is_invariant_under_rebatching_and_sharding(stream, world):
    acts, k, bounds = stream
    n, c = acts.shape
    x = torch.from_numpy(acts).to(DEV)
    one = ActMax(n_collect=k, n_latents=c, tie_mode="total")
    one.update(x, torch.arange(n))
    split = ActMax(n_collect=k, n_latents=c, tie_mode="total")
    for s, e in zip(bounds[:-1], bounds[1:]):
        if e > s:
            split.update(x[s:e], torch.arange(s, e))
    assert np.array_equal(bits(split.activations), bits(one.activations))
    assert np.array_equal(split.sample_ids.numpy(), one.sample_ids.numpy())
    # contiguous shards, merged with K4 on shard 0 (empty shards contribute the initial state)
    edges = [n * r // world for r in range(world + 1)]
    shards = []
    for r in range(world):
        am = ActMax(n_collect=k, n_latents=c, tie_mode="total")
        if edges[r + 1] > edges[r]:
            am.update(x[edges[r]:edges[r + 1]], torch.arange(edges[r], edges[r + 1]))
        else:
            am._setup_tensors()
        shards.append(am)
    ov = torch.stack([s_.device_state(DEV)[0] for s_ in shards[1:]])
    oi = torch.stack([s_.device_state(DEV)[1] for s_ in shards[1:]])
    shards[0].merge_states(ov, oi)
    assert np.array_equal(bits(shards[0].activations), bits(one.activations))
    assert np.array_equal(shards[0].sample_ids.numpy(), one.sample_ids.numpy())
