"""BASELINE.json configurations as parity cases (the bench line is configs[1]; the others are checked here).

configs[0]: ResNet-18 (random init), 1k synthetic 224x224 images, layer4 only, CLIP RN50 embed (random init, NativeClip), CPU
            reference path — top-k indices and the concept_db tensor are checked against the CPU restatement fed with the same
            layer4 activations and the embeddings of the calls that built it.
configs[3]: ViT-B/16-shaped probed model, encoder-block outputs (B,197,768), token aggregators.
configs[4]: ConvNeXt-L stage shapes (192x56x56 ... 1536x7x7) through K1 + the full-db polysemanticity score.
"""
import numpy as np
import pytest
import torch

import oracle
import synth
from semanticlens_amd import Lens, scores
from semanticlens_amd import _native as N
from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators
from semanticlens_amd.component_visualization.activation_caching import ActMaxCache

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


class _DeviceImages(torch.utils.data.Dataset):
    """synthetic images generated in one go (the per-item generator of synth.SyntheticImageDataset is slow for 1k)."""

    def __init__(self, n, mode):
        u8 = synth.synth_images_u8(torch.arange(n, device=DEV)).cpu()
        self.mode, self.u8, self.name = mode, u8, f"synthetic-{n}"

    def __len__(self):
        return self.u8.shape[0]

    def __getitem__(self, i):
        if self.mode == "model":
            return synth.normalize_u8(self.u8[i : i + 1], synth.IMAGENET_MEAN, synth.IMAGENET_STD)[0], 0
        return self.u8[i]


class _Recording:
    """An ``AbstractVLM`` that hands everything to ``fm`` and keeps the image embeddings it produced, in call order — the
    convolution trunk of a CLIP-ResNet runs on MIOpen, whose algorithm choice below B = 256 is not run-to-run deterministic, so
    the concept_db is checked against the embeddings of the very calls that built it."""

    def __init__(self, fm):
        self.fm, self.name, self.embeds = fm, getattr(fm, "name", "recording"), []

    device = property(lambda self: self.fm.device)

    def to(self, device):
        return self.fm.to(device)

    def encode_image(self, x):
        out = self.fm.encode_image(x)
        self.embeds.append(out.detach().float().cpu())
        return out

    def encode_text(self, t):
        return self.fm.encode_text(t)

    def preprocess(self, x):
        return self.fm.preprocess(x)

    def tokenize(self, t, *a, **k):
        return self.fm.tokenize(t, *a, **k)


def test_config0_resnet18_1k_images_layer4_clip_rn50_embed_vs_cpu_path():
    """configs[0] with its NAMED embed model: CLIP RN50 (random init; ModifiedResNet trunk on PyTorch, attention pool + projection
    + text tower on the kernels through NativeClip), ResNet-18 `layer4`, 1 000 images."""
    from semanticlens_amd.foundation_models.native_clip import NativeClip

    n, k, B = 1000, 20, 64
    model = synth.resnet18().to(DEV)
    fm = _Recording(NativeClip(synth.SyntheticClipRN50(device=DEV)))
    cv = ActivationComponentVisualizer(model, _DeviceImages(n, "model"), _DeviceImages(n, "fm"), ["layer4"], num_samples=k,
                                       aggregate_fn=aggregators.aggregate_conv_max, tie_mode="aten")
    # CPU path fed with the very same layer4 activations (copied off the device as they are produced)
    ref = oracle.ActMaxOracle(k, 512, oracle.MODE_ATEN)
    seen = [0]

    def tap(m, i, o):
        a = o.detach().cpu().numpy()
        ref.update(oracle.agg_conv(a, "max"), np.arange(seen[0], seen[0] + a.shape[0]))
        seen[0] += a.shape[0]

    h = model.layer4.register_forward_hook(tap)
    lens = Lens(fm, device=DEV)
    db = lens.compute_concept_db(cv, batch_size=B)
    h.remove()
    # the embedding pass does not run the probed model, so the tap saw each image exactly once
    assert seen[0] == n
    am = cv.actmax_cache.cache["layer4"]
    assert np.array_equal(bits(am.activations), ref.vals)
    assert np.array_equal(am.sample_ids.numpy(), ref.ids)  # top-k indices, ties included
    # concept_db tensor = embeds[ids] on the CPU path
    emb = torch.cat(fm.embeds).numpy()
    assert emb.shape == (n, 1024) and db["layer4"].shape == (512, k, 1024)
    assert np.array_equal(db["layer4"].numpy(), oracle.gather_rows(emb, ref.ids))
    assert len(am.alive_latents) > 0
    # text probing in RN50-CLIP's joint space against the oracle
    probes = lens.text_probing(["a striped zebra", "car wheel", "sky"], db["layer4"].mean(1))
    q = fm.encode_text(fm.tokenize(["a striped zebra", "car wheel", "sky"])).float().cpu().numpy()
    np.testing.assert_allclose(probes.cpu().numpy(), oracle.similarity(q, db["layer4"].mean(1).numpy()), rtol=0, atol=1e-4)


def test_config3_vit_b16_block_outputs_token_aggregators():
    """12 encoder blocks of width 768 over 197 tokens (ViT-B/16 geometry, 2 blocks here), max and CLS-token aggregation."""
    torch.manual_seed(0)
    blocks = torch.nn.ModuleList([synth._Block(768, 12) for _ in range(2)]).to(DEV).eval()

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = blocks

        def forward(self, x):
            for b in self.blocks:
                x = b(x)
            return x

    model = Enc()
    for fn, name, pos in ((aggregators.aggregate_transformer_max, "max", 0), (aggregators.get_aggregate_transformer_special_token(0), "token", 0)):
        cache = ActMaxCache(["blocks.0", "blocks.1"], fn, n_collect=20, tie_mode="aten")
        refs = {n: oracle.ActMaxOracle(20, 768, oracle.MODE_ATEN) for n in cache.layer_names}
        count = {n: 0 for n in refs}

        def tap(name_):
            def f(m, i, o):
                a = o.detach().cpu().numpy()
                refs[name_].update(oracle.agg_tokens(a, name, pos), np.arange(count[name_], count[name_] + a.shape[0]))
                count[name_] += a.shape[0]
            return f

        hs = [model.blocks[i].register_forward_hook(tap(f"blocks.{i}")) for i in range(2)]
        g = torch.Generator(device=DEV).manual_seed(1)
        with torch.no_grad(), cache.hook_context(model):
            for _ in range(5):
                model(torch.randn(48, 197, 768, device=DEV, generator=g))
        for h in hs:
            h.remove()
        for n_ in refs:
            assert np.array_equal(bits(cache.cache[n_].activations), refs[n_].vals), (name, n_)
            assert np.array_equal(cache.cache[n_].sample_ids.numpy(), refs[n_].ids), (name, n_)


@pytest.mark.parametrize("shape", [(32, 192, 56, 56), (32, 384, 28, 28), (32, 768, 14, 14), (32, 1536, 7, 7)])
def test_config4_convnext_l_stage_shapes(shape):
    g = torch.Generator(device=DEV).manual_seed(shape[1])
    x = torch.randn(*shape, device=DEV, generator=g)
    out = torch.empty(shape[:2], dtype=torch.float32, device=DEV)
    cand = torch.empty(shape[:2], dtype=torch.bfloat16, device=DEV)
    N.reduce_conv(x, N.SL_CONV_MAX, cand, out)
    assert torch.equal(out, x.flatten(2).amax(-1))
    assert np.array_equal(out[:4].cpu().numpy(), oracle.agg_conv(x[:4].cpu().numpy(), "max"))
    assert torch.equal(cand, out.to(torch.bfloat16))


def test_config4_polysemanticity_over_a_full_concept_db():
    """eval_polysemanticity over every component of a ConvNeXt-L-sized layer (1536 components, k=20, D=512)."""
    rng = np.random.RandomState(4)
    V = rng.randn(1536, 20, 512).astype(np.float32)
    V[::7, :9] += 2.0 * rng.randn(220, 1, 512).astype(np.float32)  # some clustered components
    got = Lens(synth.SyntheticClip(device=DEV, v_layers=1, t_layers=1), device=DEV).eval_polysemanticity({"stage4": torch.from_numpy(V).to(DEV)})["stage4"]
    assert got.shape == (1536,) and got.dtype == torch.float64
    from test_gpu_parity import assert_polysemanticity_matches

    sub = np.arange(0, 1536, 3)  # 512 components through scikit-learn (~4 s); every one must match
    assert_polysemanticity_matches(got.cpu().numpy()[sub], V[sub], "config4")


# ---- configs[3] and configs[4] on the REAL models (full depth, full width), at a size the oracle covers ----------
def _tap_oracle(model, layers, k, agg):
    """Forward hooks that stream the hooked layers' outputs through the oracle (aggregate + ActMax.update, ATen order)."""
    refs, seen, handles = {}, {n: 0 for n in layers}, []
    modules = dict(model.named_modules())

    def tap(name):
        def fn(m, i, o):
            a = agg(o.detach().float().cpu().numpy())
            if name not in refs:
                refs[name] = oracle.ActMaxOracle(k, a.shape[1], oracle.MODE_ATEN)
            refs[name].update(a, np.arange(seen[name], seen[name] + a.shape[0]))
            seen[name] += a.shape[0]

        return fn

    for n in layers:
        handles.append(modules[n].register_forward_hook(tap(n)))
    return refs, seen, handles


def test_config3_full_geometry_vit_b16_x12_so400m_embed_text_probing():
    """configs[3]: ViT-B/16 probed model, ALL 12 encoder blocks (197 x 768), SigLIP-so400m embed at FULL depth (27 x 1152, MLP
    4304, 256 tokens; NativeSigLip), text_probing through the so400m text tower (D = 1152) — top-k bits and ids, concept_db and
    probing scores against the oracle."""
    from semanticlens_amd.foundation_models import NativeSigLip

    n, k, B = 96, 5, 32
    vit = synth.vit_b16().to(DEV)
    fm = NativeSigLip(synth.SyntheticSigLip(device=DEV))
    layers = [f"blocks.{i}" for i in range(12)]
    cv = ActivationComponentVisualizer(vit, _DeviceImages(n, "model"), _DeviceImages(n, "fm"), layers, num_samples=k,
                                       aggregate_fn=aggregators.aggregate_transformer_max, tie_mode="aten")
    refs, seen, handles = _tap_oracle(vit, layers, k, lambda a: oracle.agg_tokens(a, "max"))
    lens = Lens(fm, device=DEV)
    db = lens.compute_concept_db(cv, batch_size=B)
    for h in handles:
        h.remove()
    assert all(v == n for v in seen.values())
    u8 = cv.dataset_fm.u8
    emb = torch.cat([fm.encode_image(fm.preprocess(u8[s : s + B])).cpu() for s in range(0, n, B)]).numpy()
    assert emb.shape == (n, 1152)
    for name in layers:
        am = cv.actmax_cache.cache[name]
        assert np.array_equal(bits(am.activations), refs[name].vals), name
        assert np.array_equal(am.sample_ids.numpy(), refs[name].ids), name
        assert db[name].shape == (768, k, 1152)
        assert np.array_equal(db[name].numpy(), oracle.gather_rows(emb, refs[name].ids)), name
    prompts = [f"a photo of a {w} {i}" for i, w in enumerate(["zebra", "wheel", "sky", "dog", "cat", "face", "text", "grass"] * 8)]
    agg_db = {name: v.mean(1) for name, v in db.items()}
    got = lens.text_probing(prompts, agg_db, templates=["a photo of a {}", "an image of {}"], batch_size=16)
    q = fm.encode_text(fm.tokenize([t.format(p) for t in ["a photo of a {}", "an image of {}"] for p in prompts])).float().cpu().numpy()
    q0 = fm.encode_text(fm.tokenize([t.format("") for t in ["a photo of a {}", "an image of {}"]])).float().cpu().numpy()
    qe = oracle.template_mean(q, q0, len(prompts))
    for name in (layers[0], layers[5], layers[11]):
        want = oracle.similarity(qe, agg_db[name].numpy())
        assert got[name].shape == (len(prompts), 768)
        np.testing.assert_allclose(got[name].cpu().numpy(), want, rtol=0, atol=1e-4)


def test_config4_full_geometry_convnext_l_collect_relevance_and_scores(monkeypatch):
    """configs[4]: the real ConvNeXt-L (198 M parameters, random init): the four stage outputs through (i) the activation collect
    against the oracle, (ii) the relevance visualizer (EpsilonPlusFlat LRP backward in PyTorch; K1 sum + abs-norm + K3 against the
    oracle fed the same relevance tensors), (iii) eval_clarity / eval_redundancy / eval_polysemanticity over the whole concept_db."""
    from semanticlens_amd.component_visualization import RelevanceComponentVisualizer
    from semanticlens_amd.component_visualization.activation_caching import ActMax

    n, k, B = 64, 6, 32
    model = synth.convnext_l().to(DEV)
    layers = [f"stages.{i}" for i in range(4)]
    widths = (192, 384, 768, 1536)
    fm = synth.SyntheticClip(device=DEV, v_layers=2, t_layers=1)
    ds_m, ds_f = _DeviceImages(n, "model"), _DeviceImages(n, "fm")
    # (i) activation collect + concept_db
    cv = ActivationComponentVisualizer(model, ds_m, ds_f, layers, num_samples=k, aggregate_fn=aggregators.aggregate_conv_max, tie_mode="aten")
    refs, seen, handles = _tap_oracle(model, layers, k, lambda a: oracle.agg_conv(a, "max"))
    lens = Lens(fm, device=DEV)
    db = lens.compute_concept_db(cv, batch_size=B)
    for h in handles:
        h.remove()
    emb = torch.cat([fm.encode_image(fm.preprocess(ds_f.u8[s : s + B])).cpu() for s in range(0, n, B)]).numpy()
    for name, c in zip(layers, widths):
        am = cv.actmax_cache.cache[name]
        assert np.array_equal(bits(am.activations), refs[name].vals), name
        assert np.array_equal(am.sample_ids.numpy(), refs[name].ids), name
        assert db[name].shape == (c, k, 512) and np.array_equal(db[name].numpy(), oracle.gather_rows(emb, refs[name].ids)), name
    # (ii) relevance visualizer on the same model: capture what the attribution produced and what reached K3
    from semanticlens_amd.component_visualization import lrp

    captured = []  # per batch: {layer: (activation, relevance)} as produced on the device

    def tapped(model_, modules, images, targets):
        res = lrp.lrp_epsilon_plus_flat(model_, modules, images, targets, epsilon=0.1, norm_pass=True)  # zennit's 1e-6 overflows here, see below
        captured.append({k_: (a.detach().float().cpu().numpy(), r.detach().float().cpu().numpy()) for k_, (a, r) in res.items()})
        return res

    fed = {}  # id(ActMax) -> list of (values (B, C) fp32, ids) handed to ActMax.update
    real_update = ActMax.update

    def spy(self, acts, sample_ids):
        fed.setdefault(id(self), []).append((acts.detach().float().cpu().numpy().copy(), np.asarray(sample_ids).copy()))
        return real_update(self, acts, sample_ids)

    monkeypatch.setattr(ActMax, "update", spy)
    n_rel, b_rel = 32, 16
    cvr = RelevanceComponentVisualizer(model, _DeviceImages(n_rel, "model"), _DeviceImages(n_rel, "fm"), layers, num_samples=k,
                                       attribution=tapped, tie_mode="aten", device=DEV)
    cvr.run(batch_size=b_rel)
    monkeypatch.undo()
    assert len(captured) == n_rel // b_rel
    for name, c in zip(layers, widths):
        for cache, idx, norm in ((cvr.actmax_cache, 1, True), (cvr.activation_cache, 0, False)):
            am = cache.cache[name]
            ref = oracle.ActMaxOracle(k, c, oracle.MODE_ATEN, init_value=-np.inf)
            for bi, (vals, ids) in enumerate(fed[id(am)]):
                # K1 sum (+ abs-norm) against the oracle on the SAME tensors: fp32 summation order differs -> 1e-5 of the row scale
                assert np.isfinite(captured[bi][name][idx]).all(), (name, idx, "the attribution produced inf / NaN")
                want = oracle.agg_conv(captured[bi][name][idx], "sum")
                if norm:
                    want = oracle.abs_norm_rows(want)
                scale = np.abs(want).max() + 1e-30
                assert np.abs(vals - want).max() <= 2e-5 * scale, (name, idx, np.abs(vals - want).max() / scale)
                ref.update(vals, ids)  # K3 on exactly the values the device merged: bit-exact
            assert np.array_equal(bits(am.activations), ref.vals), (name, idx)
            assert np.array_equal(am.sample_ids.numpy(), ref.ids), (name, idx)
    # relevance really flowed into every stage (a dead LRP backward would leave zeros everywhere)
    assert all(np.abs(captured[0][name][1]).max() > 0 for name in layers)
    # with zennit's default stabiliser (1e-6) the epsilon rule overflows on this architecture: reported, not collected
    cv_bad = RelevanceComponentVisualizer(model, _DeviceImages(16, "model"), _DeviceImages(16, "fm"), layers, num_samples=k, device=DEV)
    with pytest.raises(FloatingPointError, match="epsilon"):
        cv_bad.run(batch_size=16)
    # ... and `epsilon=` on the visualizer is the way around it
    cv_ok = RelevanceComponentVisualizer(model, _DeviceImages(16, "model"), _DeviceImages(16, "fm"), layers, num_samples=k, device=DEV, epsilon=0.1,
                                         composite="epsilon_plus_flat_normpass")
    cv_ok.run(batch_size=16)
    assert cv_ok.composite == "lrp_epsilon_plus_flat_normpass_eps0.1" and all(int(cv_ok.get_max_reference(n).max()) < 16 for n in layers)
    # (iii) scores over the whole concept_db
    dev_db = {name: v.to(DEV) for name, v in db.items()}
    cl, po = lens.eval_clarity(dev_db), lens.eval_polysemanticity(dev_db)
    rd = lens.eval_redundancy({name: v.mean(1) for name, v in dev_db.items()})
    for name in layers:
        V = db[name].numpy()
        np.testing.assert_allclose(cl[name].cpu().numpy(), oracle.clarity(V), rtol=0, atol=1e-5)
        assert abs(float(rd[name]) - float(oracle.redundancy(V.mean(1)))) < 1e-5
        sub = np.arange(0, V.shape[0], max(1, V.shape[0] // 24))[:24]
        np.testing.assert_allclose(po[name].cpu().numpy()[sub], oracle.polysemanticity(V[sub]), rtol=0, atol=1e-5)
