"""BASELINE configs[1] end to end under pytest: ResNet-50 (random init) layer2-4 + CLIP ViT-B/32 (random init, native towers)
through `Lens.compute_concept_db` — the reference's call (lens.py:278-329 -> activation_based.py:360-390) — over 768 synthetic
images at the bench batch size, in both tie modes, against the CPU oracle:

* top-k values and sample ids of every layer bit-equal to the oracle's aggregate + ActMax restatement fed the SAME device
  activations (streamed out of the forward by tap hooks);
* the returned concept_db equal to the oracle's gather of the embedding table;
* the native encoder's embeddings within 1e-4 of the torch module's (north_star tolerance).

`bench.py::self_check` runs the same comparison on the bench's own loop; this is the API path with host `Dataset`s."""
import numpy as np
import pytest
import torch

import oracle
import synth
from semanticlens_amd import Lens
from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators
from semanticlens_amd.foundation_models.native_clip import NativeClip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LAYERS = ["layer2", "layer3", "layer4"]
WIDTHS = {"layer2": 512, "layer3": 1024, "layer4": 2048}
N_IMAGES, BATCH, K = 768, 256, 20


@pytest.fixture(scope="module")
def world():
    model = synth.resnet50().to(DEV)
    base = synth.SyntheticClip(device=DEV)
    u8 = torch.cat([synth.synth_images_u8(torch.arange(s, s + BATCH, device=DEV)).cpu() for s in range(0, N_IMAGES, BATCH)])

    class DS(torch.utils.data.Dataset):
        def __init__(self, mode):
            self.mode, self.name = mode, f"configs1-{N_IMAGES}"

        def __len__(self):
            return N_IMAGES

        def __getitem__(self, i):
            if self.mode == "model":
                return synth.normalize_u8(u8[i:i + 1], synth.IMAGENET_MEAN, synth.IMAGENET_STD)[0], 0
            return u8[i]

    return model, base, DS


@pytest.mark.parametrize("single_pass", [False, True])
@pytest.mark.parametrize("tie_mode", ["aten", "total"])
def test_configs1_concept_db_matches_oracle(world, tie_mode, single_pass):
    model, base, DS = world
    fm = NativeClip(base)
    cv = ActivationComponentVisualizer(model, DS("model"), DS("fm"), LAYERS, num_samples=K, aggregate_fn=aggregators.aggregate_conv_max,
                                       tie_mode=tie_mode)
    mode = oracle.MODE_ATEN if tie_mode == "aten" else oracle.MODE_TOTAL
    refs, seen = {}, {n: 0 for n in LAYERS}

    def tap(name):
        def fn(m, i, o):  # the oracle's side of the comparison, streamed: host aggregate + host top-k on the same activations
            a = oracle.agg_conv(o.detach().cpu().numpy(), "max")
            refs.setdefault(name, oracle.ActMaxOracle(K, a.shape[1], mode)).update(a, np.arange(seen[name], seen[name] + a.shape[0]))
            seen[name] += a.shape[0]

        return fn

    taps = [getattr(model, n).register_forward_hook(tap(n)) for n in LAYERS]
    try:
        db = Lens(fm, device=DEV).compute_concept_db(cv, batch_size=BATCH, single_pass=single_pass)
    finally:
        for h in taps:
            h.remove()
    assert all(seen[n] == N_IMAGES for n in LAYERS)
    embeds = cv._embed_vision_dataset(fm, BATCH).cpu().numpy()  # same batches, same kernels: the table the build gathered from
    assert embeds.shape == (N_IMAGES, 512)
    for name in LAYERS:
        am, ref = cv.actmax_cache.cache[name], refs[name]
        assert am.activations.shape == (WIDTHS[name], K)
        assert np.array_equal(am.activations.view(torch.int16).numpy().view(np.uint16), ref.vals), (name, "values")
        assert np.array_equal(am.sample_ids.numpy(), ref.ids), (name, "ids")
        assert db[name].shape == (WIDTHS[name], K, 512) and db[name].device.type == "cpu"
        assert np.array_equal(db[name].numpy(), oracle.gather_rows(embeds, ref.ids)), (name, "concept_db")
    # every stored id is a real sample (or the -1 of a never-filled slot: a dead ReLU channel's +0.0 does not beat the -0.0
    # sentinel, as in the reference) and no component lists a sample twice
    for name in LAYERS:
        ids = cv.get_max_reference(name).numpy()
        assert ids.min() >= -1 and ids.max() < N_IMAGES
        assert all(len(set(row[row >= 0])) == (row >= 0).sum() for row in ids)
        assert (ids >= 0).mean() > 0.9


def test_configs1_native_embeddings_within_tolerance_of_float64(world):
    """north_star: embedding values within 1e-4 (fp32).  NativeClip (split-bf16 x3 and fp32-MFMA GEMMs) against the SAME weights
    run in float64 on the device, absolute bar on the un-normalised ViT-B/32 features of a full 256-image batch (VERDICT r03 #8:
    the round-3 test compared with the fp32 torch module, relative to the feature scale)."""
    import copy

    model, base, DS = world
    ds = DS("fm")
    u8 = torch.stack([ds[i] for i in range(BATCH)]).to(DEV)
    x = base.preprocess(u8)
    with torch.no_grad():
        want = copy.deepcopy(base.model).double().encode_image(x.double())
    for gemm in ("bf16x3", "f32"):
        got = NativeClip(base, gemm=gemm).encode_image(x)
        worst = (got.double() - want).abs().max().item()
        assert worst < 1e-4, (gemm, worst, want.abs().max().item())
        cos = torch.nn.functional.cosine_similarity(got.double(), want, dim=-1)
        assert (1 - cos).abs().max().item() < 1e-9, gemm


def test_small_loader_batches_are_embedded_in_large_ones_with_the_same_bits(world):
    """With a native foundation model the embed stage holds preprocessed loader batches back until ``fm.embed_accumulate``
    (256) images are there — the reference's default ``batch_size=32`` would otherwise run the encoder at an eighth of its
    batch.  An embedding does not depend on the batch it is computed in, so the table is bit-identical to the walk that
    encodes batch by batch (accumulation off) and to the 256-image walk; loader batches of 40 divide neither 256 nor the
    768 images (the last encoder call is a partial one).  The single-pass build gathers from that same table.  (The
    probed model's activations are PyTorch's and DO depend on the batch size — MIOpen picks its algorithms per shape —,
    so the comparison is on the embed stage, not on the top-k of two batch sizes.)"""
    model, base, DS = world
    fm = NativeClip(base)
    cv = ActivationComponentVisualizer(model, DS("model"), DS("fm"), LAYERS, num_samples=K, aggregate_fn=aggregators.aggregate_conv_max,
                                       tie_mode="total")
    calls = []
    orig = fm.encode_image
    fm.encode_image = lambda x: (calls.append(int(x.shape[0])), orig(x))[1]
    try:
        fm.embed_accumulate = 0
        ref = cv._embed_vision_dataset(fm, 256)
        assert calls == [256, 256, 256]
        calls.clear()
        one = cv._embed_vision_dataset(fm, 40)
        assert calls == [40] * (N_IMAGES // 40) + [N_IMAGES % 40]
        calls.clear()
        fm.embed_accumulate = 256
        acc = cv._embed_vision_dataset(fm, 40)
        assert calls == [280, 280, 208]  # seven loader batches of 40 reach 256
        subset = list(range(5, 700, 3))
        calls.clear()
        sub = cv._embed_vision_dataset(fm, 40, subset=subset)
        assert calls == [len(subset)]  # 232 images: below the threshold, one call at the end
        assert torch.equal(ref, one) and torch.equal(ref, acc) and torch.equal(ref[subset], sub)
        calls.clear()
        db = Lens(fm, device=DEV).compute_concept_db(cv, batch_size=40, single_pass=True)
        assert calls == [280, 280, 208]
        for name in LAYERS:
            ids = cv.get_max_reference(name)
            assert torch.equal(db[name], ref.cpu()[ids]), name
    finally:
        del fm.encode_image
        fm.embed_accumulate = type(fm).embed_accumulate
