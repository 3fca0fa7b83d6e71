"""Edge cases of the device-resident ActMax plumbing (ring growth, k above one wave, mixed batch sizes,
state moves, flush-on-read)."""
import numpy as np
import pytest
import torch

import oracle
from semanticlens_amd.component_visualization import aggregators as agg
from semanticlens_amd.component_visualization.activation_caching import ActMax, ActMaxCache

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("mode", ["total", "aten"])
def test_growing_and_shrinking_batches_through_hooks(mode, monkeypatch):
    monkeypatch.setenv("SEMANTICLENS_AMD_MERGE_EVERY", "3")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 24, 3, padding=1), torch.nn.ReLU()).to(DEV).eval()
    cache = ActMaxCache(["1"], agg.aggregate_conv_max, n_collect=70, tie_mode=mode)  # k > 64: two lane chunks
    ref = oracle.ActMaxOracle(70, 24, oracle.MODE_TOTAL if mode == "total" else oracle.MODE_ATEN)
    grabbed = []
    h = model[1].register_forward_hook(lambda m, i, o: grabbed.append(o.detach().cpu().numpy()))
    g = torch.Generator().manual_seed(1)
    start = 0
    with torch.no_grad(), cache.hook_context(model):
        for b in (4, 4, 9, 1, 33, 2, 2, 2, 65, 7):  # ring must grow when a larger batch arrives
            model(torch.randn(b, 3, 10, 10, generator=g).to(DEV))
            ref.update(oracle.agg_conv(grabbed[-1], "max"), np.arange(start, start + b))
            start += b
            if b == 33:  # reading mid-stream flushes queued merges and must not disturb what follows
                assert np.array_equal(bits(cache.cache["1"].activations), ref.vals)
    h.remove()
    assert np.array_equal(bits(cache.cache["1"].activations), ref.vals)
    assert np.array_equal(cache.cache["1"].sample_ids.numpy(), ref.ids)


def test_state_survives_load_then_continue(tmp_path):
    """Resume: store -> load into a fresh ActMax -> keep collecting == one uninterrupted stream."""
    rng = np.random.RandomState(0)
    acts = np.maximum(rng.randn(300, 40), 0).astype(np.float32)
    whole = ActMax(12, 40, tie_mode="total")
    whole.update(torch.from_numpy(acts), torch.arange(300))
    first = ActMax(12, 40, tie_mode="total")
    first.update(torch.from_numpy(acts[:170]), torch.arange(170))
    first.store(tmp_path / "s.safetensors", metadata={"n_collect": "12", "n_latents": "40"})
    resumed = ActMax.load(tmp_path / "s.safetensors")
    resumed.tie_mode = "total"
    resumed.update(torch.from_numpy(acts[170:]), torch.arange(170, 300))
    assert np.array_equal(bits(resumed.activations), bits(whole.activations))
    assert torch.equal(resumed.sample_ids, whole.sample_ids)


def test_large_k_and_limits():
    rng = np.random.RandomState(1)
    acts = rng.randn(700, 6).astype(np.float32)
    for mode, omode in (("total", oracle.MODE_TOTAL), ("aten", oracle.MODE_ATEN)):
        am = ActMax(600, 6, tie_mode=mode)
        ref = oracle.ActMaxOracle(600, 6, omode)
        for s in range(0, 700, 128):
            e = min(700, s + 128)
            am.update(torch.from_numpy(acts[s:e]), torch.arange(s, e))
            ref.update(acts[s:e], np.arange(s, e))
        assert np.array_equal(bits(am.activations), ref.vals) and np.array_equal(am.sample_ids.numpy(), ref.ids)
    with pytest.raises(ValueError, match="exceeds the supported maximum"):
        ActMax(5000, 4, tie_mode="total").update(torch.randn(8, 4), torch.arange(8))
    # the k boundary: k = 2048 needs 96 KiB of dynamic LDS per workgroup (above the 64 KiB default limit)
    big = rng.randn(2100, 3).astype(np.float32)
    am = ActMax(2048, 3, tie_mode="total")
    ref = oracle.ActMaxOracle(2048, 3, oracle.MODE_TOTAL)
    for s in range(0, 2100, 700):
        am.update(torch.from_numpy(big[s:s + 700]), torch.arange(s, s + 700))
        ref.update(big[s:s + 700], np.arange(s, s + 700))
    assert np.array_equal(bits(am.activations), ref.vals) and np.array_equal(am.sample_ids.numpy(), ref.ids)
    other = ActMax(2048, 3, tie_mode="total")
    other.update(torch.from_numpy(big[:100] + 1), torch.arange(5000, 5100))
    am.merge_states(other.device_state()[0][None], other.device_state()[1][None])  # K4 at the boundary
    assert am.activations.shape == (3, 2048)


def test_half_precision_activations_through_hooks():
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3), torch.nn.ReLU()).to(DEV).half().eval()
    cache = ActMaxCache(["1"], agg.aggregate_conv_max, n_collect=5, tie_mode="aten")
    ref = oracle.ActMaxOracle(5, 16, oracle.MODE_ATEN)
    grabbed = []
    h = model[1].register_forward_hook(lambda m, i, o: grabbed.append(o.detach().float().cpu().numpy()))
    g = torch.Generator().manual_seed(2)
    with torch.no_grad(), cache.hook_context(model):
        for i in range(3):
            model(torch.randn(8, 3, 9, 9, generator=g).to(DEV).half())
            ref.update(oracle.agg_conv(grabbed[-1], "max"), np.arange(8 * i, 8 * i + 8))
    h.remove()
    assert np.array_equal(bits(cache.cache["1"].activations), ref.vals)
    assert np.array_equal(cache.cache["1"].sample_ids.numpy(), ref.ids)


def test_tuple_outputs_raise_like_the_reference():
    class Two(torch.nn.Module):
        def forward(self, x):
            return x, x

    model = torch.nn.Sequential(Two()).to(DEV)
    cache = ActMaxCache(["0"], agg.aggregate_conv_max, n_collect=3)
    with pytest.raises(AttributeError), cache.hook_context(model):
        model(torch.randn(2, 3, 4, 4, device=DEV))


def test_reduce_policy_tuner_settles_per_layer_and_changes_no_result(monkeypatch):
    """The per-layer cache-policy tuner of the collect hooks (`N.ReducePolicyTuner`): a 155 MB token activation is reduced under
    both candidate policies on its first launches, a choice is made without any host synchronisation, and the top-k state equals
    the one collected with the tuner off, bit for bit."""
    import numpy as np
    import torch

    from semanticlens_amd import _native as N
    from semanticlens_amd.component_visualization import aggregators
    from semanticlens_amd.component_visualization.activation_caching import ActMaxCache

    class Add(torch.nn.Module):  # output = residual add of two other tensors, like a transformer block
        def forward(self, x):
            return x + 0.25 * x.flip(1)

    def collect(autotune):
        monkeypatch.setenv("SL_REDUCE_AUTOTUNE", "1" if autotune else "0")
        N.ReducePolicyTuner._registry.clear()
        model = torch.nn.Sequential(Add()).to("cuda:0")
        cache = ActMaxCache(["0"], aggregators.aggregate_transformer_max, n_collect=7, tie_mode="aten")
        g = torch.Generator(device="cuda:0").manual_seed(3)
        with torch.no_grad(), cache.hook_context(model):
            for _ in range(14):  # one untimed + three timed launches per candidate policy
                model(torch.randn(256, 197, 768, device="cuda:0", generator=g))
        torch.cuda.synchronize()
        am = cache.cache["0"]
        if am._policy_tuner is not None and autotune:
            am._policy_tuner._harvest()  # in a real run the next launches do this; here the host finished enqueuing before the device ran
        return am.activations.view(torch.int16).numpy().copy(), am.sample_ids.numpy().copy(), am._policy_tuner

    v1, i1, t1 = collect(True)
    n_cand = len(N.ReducePolicyTuner.CANDIDATES)
    assert t1.choice in range(n_cand) and len(t1.medians_ns_per_mb) == n_cand and all(m > 0 for m in t1.medians_ns_per_mb)
    v0, i0, t0 = collect(False)
    assert t0.choice is None
    assert np.array_equal(v1, v0) and np.array_equal(i1, i0)
    # an explicit policy keeps the tuner out
    N.set_reduce_policy(0, 0)
    try:
        _, _, t2 = collect(True)
        assert t2.choice is None
    finally:
        N.set_reduce_policy(None, None)


@pytest.mark.parametrize("seed,nw", [(11, None), (12, "8")])
def test_colreduce2_fuzz_against_the_oracle(seed, nw):
    """`tools/fuzz_k2.py` as a test: 150 random (B, T, F) token tensors and channels_last maps per seed — F from one 16-byte piece to
    4 352 components, 1-700 rows, fp32 / fp16 / bf16, every aggregator, planted NaN / +-inf / -0.0 — through colreduce2 (every LPR /
    wave-count variant; `SL_OPTIONS=colreduce_nw=8` forces the eight-wave instances) against the oracle, synchronising after every call."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if nw:
        env["SL_OPTIONS"] = f"colreduce_nw={nw}"
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_k2.py"), str(seed), "150"], env=env, capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert res.returncode == 0 and "fuzz_k2 ok: 150 cases" in res.stdout, (res.stdout[-600:], res.stderr[-1500:])


@pytest.mark.parametrize("B,c,k", [(6000, 5, 20), (3000, 9, 100), (15000, 3, 20), (1279, 40, 20), (1281, 40, 20), (700, 300, 7)])
def test_aten_update_long_rows_and_both_branches(B, c, k):
    """K3 in the reference's tie order at row lengths that change the kernel's shape: rows of tens of KB in LDS (one or two rows per
    workgroup, more than 64 KB of dynamic LDS), rows too long for the LDS (the one-lane-per-row kernel), and batch sizes on both
    sides of torch.topk's `k * 64 <= n` switch between partial_sort and nth_element + sort."""
    rng = np.random.RandomState(B + c + k)
    am = ActMax(k, c, tie_mode="aten")
    ref = oracle.ActMaxOracle(k, c, oracle.MODE_ATEN)
    for step in range(2):
        acts = (rng.randint(0, 40, size=(B, c)).astype(np.float32) / 8) if step else np.maximum(rng.randn(B, c), 0).astype(np.float32)
        ids = np.arange(step * B, (step + 1) * B)
        am.update(torch.from_numpy(acts), torch.from_numpy(ids))
        ref.update(acts, ids)
        assert np.array_equal(bits(am.activations), ref.vals), step
        assert np.array_equal(am.sample_ids.numpy(), ref.ids), step
