"""End-to-end drop-in check on the GPU: Lens.compute_concept_db / text_probing / eval_* of the build
against outputs of the unmodified reference on the same (integer-valued, hence order-independent) model
and data — tests/golden/pipeline.npz, produced by tests/golden/make_golden.py."""
import numpy as np
import pytest
import torch
from safetensors import safe_open

from helpers import FakeVLM, TensorPairDataset, make_int_conv_model, make_int_images
from semanticlens_amd import Lens
from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def _build(tmp, tie_mode):
    model = make_int_conv_model().to(DEV)
    ds = TensorPairDataset(make_int_images(40))
    cv = ActivationComponentVisualizer(
        model, ds, ds, layer_names=["0", "2"], num_samples=6, aggregate_fn=aggregators.aggregate_conv_max,
        cache_dir=tmp, tie_mode=tie_mode,
    )
    fm = FakeVLM().to(DEV)
    return cv, Lens(fm, device=DEV)


def test_pipeline_matches_reference_bit_exact_in_aten_mode(golden, tmp_path):
    g = golden("pipeline")
    cv, lens = _build(str(tmp_path), "aten")
    db = lens.compute_concept_db(cv, batch_size=16)
    for name in ("0", "2"):
        am = cv.actmax_cache.cache[name]
        assert np.array_equal(bits(am.activations), g[f"vals_{name}"])
        assert np.array_equal(am.sample_ids.numpy(), g[f"ids_{name}"])  # top-k indices bit-exact
        assert np.array_equal(cv.get_max_reference(name).numpy(), g[f"ids_{name}"])
        assert db[name].device.type == "cpu" and db[name].dtype == torch.float32
        assert np.array_equal(db[name].numpy(), g[f"db_{name}"])  # concept_db tensor
    agg_db = {k: v.mean(1) for k, v in db.items()}
    probe = lens.text_probing(["cat", "dog"], agg_db, templates=["a photo of a {}"])
    clar = lens.eval_clarity(db)
    red = lens.eval_redundancy(agg_db)
    for name in ("0", "2"):
        np.testing.assert_allclose(probe[name].numpy(), g[f"probe_{name}"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(clar[name].numpy(), g[f"clarity_{name}"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(red[name].numpy(), g[f"redundancy_{name}"], rtol=0, atol=1e-5)
    # on-disk cache layout equals the reference's: same relative paths, tensor names and metadata
    files = sorted(str(p.relative_to(tmp_path)) for p in tmp_path.rglob("*.safetensors"))
    assert files == [str(f) for f in g["cache_files"]]
    for f, want in zip(files, g["cache_meta"]):
        with safe_open(str(tmp_path / f), framework="pt") as fh:
            got = repr(sorted((fh.metadata() or {}).items())) + "|" + repr(sorted(fh.keys()))
        assert got == str(want)
    # second visualizer + lens: everything is served from the caches the first run wrote
    cv2, lens2 = _build(str(tmp_path), "aten")
    db2 = lens2.compute_concept_db(cv2, batch_size=16)
    assert np.array_equal(db2["2"].numpy(), g["db_2"])
    assert np.array_equal(cv2.get_max_reference("0").numpy(), g["ids_0"])


def test_pipeline_total_mode_same_values_valid_ids(golden, tmp_path):
    g = golden("pipeline")
    cv, lens = _build(None, "total")
    db = lens.compute_concept_db(cv, batch_size=16, keep_on_device=True)
    images = torch.from_numpy(g["images"])
    model = make_int_conv_model()
    with torch.no_grad():
        a0 = model[0](images)
        acts = {"0": a0.flatten(2).amax(-1), "2": model[2](model[1](a0)).flatten(2).amax(-1)}
    for name in ("0", "2"):
        am = cv.actmax_cache.cache[name]
        v = am.activations.float().numpy()
        assert np.array_equal(v, torch.from_numpy(g[f"vals_{name}"].view(np.int16)).view(torch.bfloat16).float().numpy())
        ids = am.sample_ids.numpy()
        for c in range(ids.shape[0]):
            real = ids[c] >= 0
            assert len(set(ids[c][real])) == real.sum()
            own = acts[name][ids[c][real], c].to(torch.bfloat16).float().numpy()
            assert np.array_equal(own, v[c][real])
        assert db[name].is_cuda
        assert np.array_equal(db[name].cpu().numpy(), g["embeds"][ids])


def test_batch_size_one_and_dataset_smaller_than_k(tmp_path):
    model = make_int_conv_model().to(DEV)
    ds = TensorPairDataset(make_int_images(5))
    cv = ActivationComponentVisualizer(model, ds, ds, ["2"], num_samples=9, aggregate_fn=aggregators.aggregate_conv_max, tie_mode="aten")
    db = Lens(FakeVLM().to(DEV)).compute_concept_db(cv, batch_size=1)
    ids = cv.get_max_reference("2")
    assert ids.shape == (16, 9) and (ids == -1).any()
    emb = FakeVLM().encode_image(torch.stack([ds[i][0] for i in range(5)])).numpy()
    assert np.array_equal(db["2"].numpy(), emb[ids.numpy()])  # -1 wraps to the LAST embedding (finding 2)


def test_zero_samples_and_empty_layers(tmp_path):
    # reference tests/component_visualization/test_activation_based.py:126-161
    model = make_int_conv_model().to(DEV)
    ds = TensorPairDataset(make_int_images(4))
    cv = ActivationComponentVisualizer(model, ds, ds, [], num_samples=10)
    assert cv.run() == {}
    cv0 = ActivationComponentVisualizer(model, ds, ds, ["0"], num_samples=0, cache_dir=str(tmp_path))
    cv0.run(batch_size=2)
    am = cv0.actmax_cache.cache["0"]
    assert am.n_collect == 0 and am.sample_ids.shape[1] == 0 and am.activations.shape[1] == 0


def test_referenced_only_embedding_gives_the_same_concept_db():
    """Embedding only the samples some component refers to == embedding the whole dataset, including
    the -1 sentinel (gathers the LAST sample) and repeated / unsorted ids; the encoder sees fewer samples."""
    model = make_int_conv_model().to(DEV)
    for n, k in ((40, 3), (5, 9)):  # plenty of unreferenced samples; fewer samples than k (-1 slots)
        ds = TensorPairDataset(make_int_images(n))
        def build():
            return ActivationComponentVisualizer(model, ds, ds, ["0", "2"], num_samples=k, aggregate_fn=aggregators.aggregate_conv_max, tie_mode="aten")
        fm_all, fm_ref = FakeVLM().to(DEV), FakeVLM().to(DEV)
        want = build()._compute_concept_db(fm_all, batch_size=4)
        cv = build()
        got = cv._compute_concept_db(fm_ref, batch_size=4, referenced_only=True)
        for name in ("0", "2"):
            assert torch.equal(got[name], want[name]), (n, k, name)
        refs = torch.cat([cv.get_max_reference(l).reshape(-1) for l in ("0", "2")])
        n_ref = len(set((refs % n).tolist()))
        assert fm_ref.calls["encode_image"] == -(-n_ref // 4) and fm_all.calls["encode_image"] == -(-n // 4)
        if n == 40:
            assert n_ref < n


def test_single_pass_two_stream_build_equals_two_pass(tmp_path):
    """collect + embed fused over one walk of the data on two HIP streams == the reference's two sequential passes:
    same top-k states, same concept DB; with a cache directory the states are stored, and a second call (cache hit)
    only embeds."""
    model = make_int_conv_model().to(DEV)
    ds = TensorPairDataset(make_int_images(37))

    def build(cache=None):
        return ActivationComponentVisualizer(model, ds, ds, ["0", "2"], num_samples=5, aggregate_fn=aggregators.aggregate_conv_max,
                                             tie_mode="aten", cache_dir=cache)

    cv_ref = build()
    want = cv_ref._compute_concept_db(FakeVLM().to(DEV), batch_size=8)
    cv = build(str(tmp_path))
    got = cv._compute_concept_db(FakeVLM().to(DEV), batch_size=8, single_pass=True)
    for name in ("0", "2"):
        assert torch.equal(cv.get_max_reference(name), cv_ref.get_max_reference(name))
        assert torch.equal(got[name], want[name])
    assert any(cv.storage_dir.rglob("*.safetensors"))
    cv2 = build(str(tmp_path))  # constructor + run() find the cache: no forward pass of the model any more
    model_calls = []
    h = model.register_forward_hook(lambda m, i, o: model_calls.append(1))
    again = cv2._compute_concept_db(FakeVLM().to(DEV), batch_size=8, single_pass=True)
    h.remove()
    assert not model_calls and all(torch.equal(again[n], want[n]) for n in ("0", "2"))


# ---- host prefetch (component_visualization/_prefetch.py) -----------------------------------------------------------
def test_prefetcher_order_exceptions_and_early_stop():
    from semanticlens_amd.component_visualization._prefetch import PinnedStack, Prefetcher, upload_stage

    data = [(torch.full((3, 4, 4), float(i)), i) for i in range(37)]
    stack = PinnedStack(slots=3)
    loader = torch.utils.data.DataLoader(data, batch_size=5, shuffle=False, collate_fn=stack)
    got = [b for b in Prefetcher(loader, upload_stage(DEV, stack), DEV, depth=2)]
    assert len(got) == 8 and all(b.is_cuda for b in got)
    assert torch.equal(torch.cat(got)[:, 0, 0, 0].cpu(), torch.arange(37, dtype=torch.float32))

    def boom(item):
        if int(item[0, 0, 0, 0]) >= 10:
            raise ValueError("bad sample")
        return item.to(DEV)

    seen = []
    with pytest.raises(ValueError, match="bad sample"):
        for b in Prefetcher(torch.utils.data.DataLoader(data, batch_size=5, collate_fn=PinnedStack()), boom, DEV):
            seen.append(b)
    assert len(seen) == 2  # the exception surfaces at its position in the stream

    pf = Prefetcher(torch.utils.data.DataLoader(data, batch_size=2, collate_fn=PinnedStack()), lambda t: t.to(DEV), DEV)
    for i, b in enumerate(pf):
        if i == 3:
            break
    pf.close()
    assert not pf._thread.is_alive()


def test_prefetch_on_and_off_build_the_same_concept_db(tmp_path):
    """The background walk changes when host work happens, nothing else: same top-k states, same concept DB, in the
    two-pass and the single-pass form."""
    x = make_int_images(70, seed=21)
    ds = TensorPairDataset(x, name="pf")
    outs = []
    for prefetch in (True, False):
        for single in (True, False):
            cv = ActivationComponentVisualizer(make_int_conv_model().to(DEV), ds, ds, ["0", "2"], num_samples=6,
                                               aggregate_fn=aggregators.aggregate_conv_max, cache_dir=None)
            cv.prefetch = prefetch
            fm = FakeVLM().to(DEV)
            db = Lens(fm, device=DEV).compute_concept_db(cv, batch_size=16, single_pass=single)
            outs.append((db, {n: cv.get_max_reference(n).clone() for n in cv.layer_names}))
    for db, ids in outs[1:]:
        for name in ids:
            assert torch.equal(ids[name], outs[0][1][name])
            assert torch.equal(db[name], outs[0][0][name])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("channels_last", [False, True])
@pytest.mark.parametrize("agg_name", ["max", "mean"])
def test_half_precision_model_through_the_hooks(dt, channels_last, agg_name):
    """A model run in fp16 / bf16 (optionally channels_last): the hooked activations arrive in that dtype and take the
    half-precision kernels.  The oracle is fed the SAME device activations (captured by a second hook), aggregated in
    fp32 and rounded once to the activation dtype for the mean, as torch's `mean` of a half tensor does."""
    import oracle

    model = make_int_conv_model().to(DEV).to(dt)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    n, k, bs = 45, 7, 16
    x = make_int_images(n, seed=13).to(dt)
    ds = TensorPairDataset(x, name="half")
    fn = aggregators.aggregate_conv_max if agg_name == "max" else aggregators.aggregate_conv_mean
    cv = ActivationComponentVisualizer(model, ds, ds, ["0", "2"], num_samples=k, aggregate_fn=fn, cache_dir=None, tie_mode="aten")
    raw = {"0": [], "2": []}
    taps = [model[int(nm)].register_forward_hook(lambda m, i, o, nm=nm: raw[nm].append(o.detach().float().cpu().numpy())) for nm in raw]
    try:
        cv.run(batch_size=bs)
    finally:
        for h in taps:
            h.remove()
    for nm, c in (("0", 8), ("2", 16)):
        ref = oracle.ActMaxOracle(k, c, oracle.MODE_ATEN)
        start = 0
        for a in raw[nm]:
            v = oracle.agg_conv(a, agg_name)
            if agg_name == "mean":
                v = torch.from_numpy(v).to(dt).float().numpy()
            ref.update(v, np.arange(start, start + a.shape[0]))
            start += a.shape[0]
        am = cv.actmax_cache.cache[nm]
        if agg_name == "max":
            assert np.array_equal(bits(am.activations), ref.vals), nm
            assert np.array_equal(am.sample_ids.numpy(), ref.ids), nm
        else:  # fp32 summation order differs: values within one ulp of the activation dtype
            got, want = oracle.bf16_to_f32(bits(am.activations)), oracle.bf16_to_f32(ref.vals)
            assert np.allclose(got, want, rtol=2.0 ** -7, atol=1e-6), nm
