"""the native towers bound to the module layouts the REFERENCE's wrappers construct.

``OpenClip(url)`` builds ``open_clip.create_model_and_transforms(url)`` (foundation_models/clip.py:52-62) — an open_clip
``CLIP`` whose image tower is ``visual`` = ``VisionTransformer`` — and ``SigLipV2()`` (clip.py:190-211,
``hf-hub:timm/ViT-B-16-SigLIP2``) an open_clip ``CustomTextCLIP`` with a timm trunk.  open_clip / timm are absent here, so
these tests drive ``NativeClip`` / ``NativeSigLip`` / ``OpenClip.native()`` with ``tests/openclip_like.py``: torch modules
with open_clip 3.0's attribute tree.  Parity = same-layout torch module (fp32, same weights) vs native, <= 1e-5 of the
feature scale in both GEMM modes (north_star: embedding values within 1e-4)."""
import sys

import numpy as np
import pytest
import torch

import openclip_like as oc
from semanticlens_amd.foundation_models.native_clip import NativeClip, NativeSigLip, native_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# fp32-MFMA GEMMs: 1e-5 of the feature scale.  split-bf16 x3 drops the lo*lo term (2^-16 per product): 1.0-1.9e-5 on these
# towers, whose weights are drawn larger than a trained model's to make every bias / gamma count; north_star allows 1e-4.
TOL = {"f32": 1e-5, "bf16x3": 3e-5}

SMALL = dict(embed_dim=64, image_size=64, patch=16, v_width=128, v_layers=2, v_heads=2, ctx=16, vocab=1000, t_width=128, t_layers=2, t_heads=2)


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


class Wrapped:
    """The members of the reference's OpenClip wrapper the native classes use: ``model``, ``to``, tokenizer / preprocess."""

    def __init__(self, model, ctx, vocab, eot):
        self.model = model.to(DEV).eval()
        self.name = type(model).__name__
        self._tok = oc._tokenizer(vocab, eot=eot)
        self._ctx = ctx

    def to(self, device):
        return self.model.to(device)

    def tokenize(self, txt, context_length=None):
        return self._tok(txt, context_length=context_length or self._ctx).to(DEV)

    def preprocess(self, img):
        return img


PROMPTS = ["a photo of a cat", "dog", "a very long prompt with many many words in it " * 3, "two red wheels on a cart"]


def check_towers(base, nat, image_size, embed_dim, tol):
    g = torch.Generator(device=DEV).manual_seed(1)
    img = torch.randn(6, 3, image_size, image_size, device=DEV, generator=g)
    with torch.no_grad():
        want_i = base.model.encode_image(img)
        tok = base.tokenize(PROMPTS)
        want_t = base.model.encode_text(tok)
    got_i, got_t = nat.encode_image(img), nat.encode_text(tok)
    assert got_i.shape == want_i.shape == (6, embed_dim) and got_i.dtype == torch.float32
    assert got_t.shape == want_t.shape == (len(PROMPTS), embed_dim)
    assert rel_err(got_i, want_i) < tol, ("image", rel_err(got_i, want_i))
    assert rel_err(got_t, want_t) < tol, ("text", rel_err(got_t, want_t))
    return got_i, got_t


VARIANTS = {
    "vit-b32-style": dict(),
    "quickgelu": dict(quick_gelu=True),  # OpenAI checkpoints ("ViT-B-32-quickgelu")
    "clipa-style": dict(pool_type="avg", no_ln_pre=True, final_ln_after_pool=True, text_pool_type="last", no_causal_mask=True),
    "avg-ln-before-pool": dict(pool_type="avg"),
    "layerscale-projbias-first": dict(ls_init_value=0.1, proj_bias=True, text_pool_type="first"),
    "head_dim80/32": dict(v_width=160, t_width=64, t_layers=1),
}


@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_native_clip_reads_the_open_clip_attribute_tree(variant, gemm):
    """`model.visual.{conv1, class_embedding, positional_embedding, ln_pre, transformer.resblocks[i].{ln_1, attn, ls_1, ln_2,
    mlp.c_fc/gelu/c_proj, ls_2}, ln_post, proj, pool_type}` and the text members on the model — the branch of
    `NativeClip.__init__` a reference user reaches through `OpenClip(url).native()`."""
    cfg = {**SMALL, **VARIANTS[variant]}
    model = oc.build_clip_vit(seed=3, **cfg)
    base = Wrapped(model, cfg["ctx"], cfg["vocab"], eot=cfg["vocab"] - 1)
    nat = NativeClip(base, gemm=gemm)
    assert type(nat.vision).__name__ == "NativeVisionTower" and nat.text is not None
    assert nat.vision.pool == cfg.get("pool_type", "tok") and nat.text.pool == cfg.get("text_pool_type", "argmax")
    assert nat.text.causal == (not cfg.get("no_causal_mask", False))
    check_towers(base, nat, cfg["image_size"], cfg["embed_dim"], TOL[gemm])
    if nat.text.pool == "argmax":  # truncation after the last end-of-text token and the pooled-row shortcut keep every bit
        tok = base.tokenize(PROMPTS[:2])
        fast = nat.encode_text(tok)
        nat.text.truncate = nat.text.pool_shortcut = False
        assert torch.equal(nat.encode_text(tok), fast)


@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
def test_open_clip_wrapper_native_full_vit_b32(monkeypatch, gemm):
    """The reference's own constructor path: `OpenClip("ViT-B-32")` (clip.py:52-62) over an importable open_clip stand-in that
    returns a full ViT-B/32-geometry CLIP (12 x 768 / 12 x 512, 77-token context), then `.native()` (clip.py:79 of the build)."""
    from semanticlens_amd.foundation_models import OpenClip

    reg = {"ViT-B-32": (lambda: oc.build_clip_vit(seed=5), 224, oc.CLIP_MEAN, oc.CLIP_STD, oc._tokenizer(49408, eot=49407))}
    fake = oc.fake_open_clip_module(reg)
    monkeypatch.setitem(sys.modules, "open_clip", fake)
    fm = OpenClip("ViT-B-32", device=DEV)
    assert fake.calls[0][:2] == ("create", "ViT-B-32") and type(fm.model).__name__ == "CLIP"
    nat = fm.native(gemm=gemm, device_preprocess=False)
    assert isinstance(nat, NativeClip) and nat.name.startswith(f"native-{gemm}-")
    g = torch.Generator(device=DEV).manual_seed(2)
    img = torch.randn(8, 3, 224, 224, device=DEV, generator=g)
    tok = fm.tokenize(PROMPTS)
    assert tok.shape == (4, 77)
    want_i, want_t = fm.encode_image(img), fm.encode_text(tok)
    got_i, got_t = nat.encode_image(img), nat.encode_text(tok)
    assert rel_err(got_i, want_i) < TOL[gemm] and rel_err(got_t, want_t) < TOL[gemm], (rel_err(got_i, want_i), rel_err(got_t, want_t))
    cos = torch.nn.functional.cosine_similarity(got_i, want_i, dim=-1)
    assert (1 - cos).abs().max().item() < 1e-6
    # host preprocessing and tokenizer stay the wrapped object's
    from PIL import Image

    pil = Image.fromarray(np.random.default_rng(0).integers(0, 256, (50, 70, 3), dtype=np.uint8))
    assert torch.equal(nat.preprocess([pil, pil]), fm.preprocess([pil, pil])) and nat.preprocess(pil).shape == (1, 3, 224, 224)


@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
@pytest.mark.parametrize("geom", [dict(embed_dim=128, image_size=64, patch=16, width=128, layers=2, heads=2, ctx=16, vocab=1000),
                                  dict(embed_dim=144, image_size=56, patch=14, width=144, layers=2, heads=2, ctx=16, vocab=1000),
                                  dict(embed_dim=96, image_size=64, patch=16, width=128, layers=1, heads=4, ctx=8, vocab=500, proj="linear",
                                       init_values=0.2)])
def test_native_siglip_reads_custom_text_clip_with_timm_trunk(geom, gemm):
    """`visual.trunk.{patch_embed.proj, pos_embed, blocks[i].{norm1, attn.qkv, attn.proj, ls1, norm2, mlp.fc1/act/fc2, ls2}, norm,
    attn_pool.{latent, q, kv, proj, norm, mlp}}`, `visual.head`, `text` = TextTransformer(no causal mask, last-token pool,
    Linear projection with bias): what `SigLipV2()` constructs (clip.py:190-211).  Second case: head_dim 72, patch 14."""
    model = oc.build_siglip2(seed=4, **geom)
    base = Wrapped(model, geom["ctx"], geom["vocab"], eot=None)
    nat = NativeSigLip(base, gemm=gemm)
    assert nat.text.pool == "last" and not nat.text.causal and nat.text.b_proj is not None
    check_towers(base, nat, geom["image_size"], geom["embed_dim"], TOL[gemm])
    assert isinstance(native_model(base, gemm=gemm), NativeSigLip)


def test_siglipv2_wrapper_native_b16_geometry(monkeypatch):
    """`SigLipV2(device).native()` end to end at the ViT-B/16-SigLIP2 geometry (12 x 768, 196 patches, 64-token context,
    256 000-word vocabulary), both towers, default GEMM mode; then through `Lens.text_probing`."""
    from semanticlens_amd import Lens
    from semanticlens_amd.foundation_models import SigLipV2

    name = "hf-hub:timm/ViT-B-16-SigLIP2"
    reg = {name: (lambda: oc.build_siglip2(seed=6), 224, oc.HALF, oc.HALF, oc._tokenizer(256000, eot=None, pad=1))}
    fake = oc.fake_open_clip_module(reg)
    monkeypatch.setitem(sys.modules, "open_clip", fake)
    fm = SigLipV2(device=DEV)
    assert fake.calls[0][:2] == ("create", name) and type(fm.model).__name__ == "CustomTextCLIP"
    nat = fm.native(device_preprocess=False)
    assert isinstance(nat, NativeSigLip)
    g = torch.Generator(device=DEV).manual_seed(3)
    img = torch.randn(8, 3, 224, 224, device=DEV, generator=g)
    tok = fm.tokenize(PROMPTS)
    assert tok.shape == (4, 64)
    want_i, want_t = fm.encode_image(img), fm.encode_text(tok)
    got_i, got_t = nat.encode_image(img), nat.encode_text(tok)
    assert rel_err(got_i, want_i) < TOL["bf16x3"] and rel_err(got_t, want_t) < TOL["bf16x3"], (rel_err(got_i, want_i), rel_err(got_t, want_t))
    db = {"layer": torch.randn(50, 768, generator=torch.Generator().manual_seed(0))}
    p_nat = Lens(nat, device=DEV).text_probing(["cat", "dog", "zebra"], db)["layer"]
    p_ref = Lens(fm, device=DEV).text_probing(["cat", "dog", "zebra"], db)["layer"]
    np.testing.assert_allclose(p_nat.cpu().numpy(), p_ref.cpu().numpy(), rtol=0, atol=1e-4)  # north_star: cosines within 1e-4


def test_open_clip_native_drops_into_the_concept_db_pipeline(monkeypatch):
    """`Lens.compute_concept_db` with `OpenClip(...).native()` as the foundation model == with the torch wrapper itself."""
    import synth
    from semanticlens_amd import Lens
    from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators
    from semanticlens_amd.foundation_models import OpenClip

    reg = {"tiny": (lambda: oc.build_clip_vit(seed=7, **SMALL), 64, oc.CLIP_MEAN, oc.CLIP_STD, oc._tokenizer(1000, eot=999))}
    monkeypatch.setitem(sys.modules, "open_clip", oc.fake_open_clip_module(reg))
    fm = OpenClip("tiny", device=DEV)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.ReLU()).to(DEV).eval()
    from PIL import Image

    u8 = synth.synth_images_u8(torch.arange(24, device=DEV), size=64).cpu()

    class DS(torch.utils.data.Dataset):
        def __init__(self, mode):
            self.mode, self.name = mode, "oc-24"

        def __len__(self):
            return 24

        def __getitem__(self, i):
            if self.mode == "model":
                return synth.normalize_u8(u8[i:i + 1], synth.IMAGENET_MEAN, synth.IMAGENET_STD)[0], 0
            return Image.fromarray(u8[i].permute(1, 2, 0).numpy())

    def build():
        return ActivationComponentVisualizer(model, DS("model"), DS("fm"), ["1"], num_samples=4, aggregate_fn=aggregators.aggregate_conv_max,
                                             tie_mode="aten")

    db_t = Lens(fm, device=DEV).compute_concept_db(build(), batch_size=8)["1"]
    db_n = Lens(fm.native(device_preprocess=False), device=DEV).compute_concept_db(build(), batch_size=8)["1"]
    assert db_n.shape == db_t.shape == (16, 4, 64)
    assert rel_err(db_n, db_t) < 1e-4


def test_unsupported_open_clip_variants_are_refused_by_name():
    """Variants the native towers do not implement raise TypeError / ValueError naming what was found, instead of running a
    different computation: CoCa attention pooling, token-sequence outputs, timm trunks with class tokens / q-k norm /
    non-MAP pooling, and towers without a ViT at all (MobileCLIP's FastViT hybrid, clip.py:214-247)."""
    def wrap(m, vocab=1000):
        return Wrapped(m, 16, vocab, eot=vocab - 1)

    with pytest.raises(TypeError, match="attn_pool"):
        NativeClip(wrap(oc.build_clip_vit(**SMALL, attentional_pool=True)))
    with pytest.raises(TypeError, match="pool_type='none'"):
        NativeClip(wrap(oc.build_clip_vit(**SMALL, pool_type="none")))
    with pytest.raises(TypeError, match="text_pool_type='none'"):
        NativeClip(wrap(oc.build_clip_vit(**SMALL, text_pool_type="none")))
    with pytest.raises(ValueError, match="head_dim"):
        NativeClip(wrap(oc.build_clip_vit(**{**SMALL, "v_width": 120, "v_heads": 6})))  # head_dim 20
    sig = dict(embed_dim=128, image_size=64, patch=16, width=128, layers=1, heads=2, ctx=16, vocab=1000)
    with pytest.raises(TypeError, match="class / register tokens"):
        NativeSigLip(wrap(oc.build_siglip2(**sig, class_token=True)))
    with pytest.raises(TypeError, match="q/k normalisation"):
        NativeSigLip(wrap(oc.build_siglip2(**sig, qk_norm=True)))
    with pytest.raises(TypeError, match="global_pool='avg'"):
        NativeSigLip(wrap(oc.build_siglip2(**sig, global_pool="avg")))

    class MobileLike(torch.nn.Module):  # a convolutional image tower behind `visual`, nothing ViT-shaped
        def __init__(self):
            super().__init__()
            self.visual = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten())

    with pytest.raises(TypeError, match="NativeClip reads open_clip's CLIP"):
        NativeClip(wrap(MobileLike()))
    with pytest.raises(TypeError, match="NativeSigLip expects"):
        NativeSigLip(wrap(MobileLike()))


# ---- MobileCLIP-style models (`ClipMobile`, clip.py:214-247: the reference tutorial's foundation model) -----------------------
class _ConvTower(torch.nn.Module):
    """A convolutional image tower behind `visual` (stands for MobileCLIP's FastViT hybrid, which the native classes do not read)."""

    def __init__(self, embed_dim):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, 2, 1), torch.nn.GELU(), torch.nn.Conv2d(16, 32, 3, 2, 1), torch.nn.GELU(),
                                        torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(32, embed_dim))

    def forward(self, x):
        return self.body(x)


def _mobile_like(layout):
    torch.manual_seed(7)
    text = oc.TextTransformer(context_length=16, vocab_size=1000, width=128, heads=2, layers=2, output_dim=64)
    if layout == "custom_text":  # open_clip.CustomTextCLIP: what MobileCLIP-S1/S2 are built as
        model = oc.CustomTextCLIP(_ConvTower(64), text)
    else:  # text members flattened onto the model, as open_clip.CLIP does
        model = oc.build_clip_vit(**SMALL)
        model.visual = _ConvTower(64)
    oc._randomize(model, 7)
    return Wrapped(model, 16, 1000, eot=999)


@pytest.mark.parametrize("layout", ["custom_text", "clip"])
@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
def test_native_text_tower_beside_a_torch_image_tower(layout, gemm):
    """`NativeTextClip`: the text tower on the kernels, the (unreadable) image tower left to PyTorch — `ClipMobile.native()`."""
    from semanticlens_amd.foundation_models import NativeTextClip
    from semanticlens_amd.foundation_models.native_clip import native_model

    base = _mobile_like(layout)
    base.encode_image = lambda img: base.model.encode_image(img)
    with pytest.raises(TypeError):
        native_model(base, gemm=gemm)  # image_tower="native": refused, as before
    nat = native_model(base, gemm=gemm, image_tower="auto")
    assert isinstance(nat, NativeTextClip) and nat.name.startswith(f"native-text-{gemm}-")
    got_i, got_t = check_towers(base, nat, 64, 64, 3e-5 if gemm == "bf16x3" else 1e-5)
    with torch.no_grad():
        assert torch.equal(got_i, base.model.encode_image(torch.randn(6, 3, 64, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))))
    # and it probes: Lens.text_probing through the native text tower
    from semanticlens_amd import Lens

    db = {"l": torch.randn(9, 64, device=DEV)}
    out = Lens(nat, device=DEV).text_probing(PROMPTS, db)
    want = torch.nn.functional.normalize(got_t, dim=-1) @ torch.nn.functional.normalize(db["l"], dim=-1).T
    assert torch.allclose(out["l"], want, atol=1e-5)


def test_clip_mobile_native_keeps_the_image_tower_on_torch(monkeypatch):
    """`ClipMobile(...).native()` — the wrapper the reference's tutorial uses — no longer refuses: `NATIVE_IMAGE_TOWER = "auto"`."""
    import sys

    from semanticlens_amd.foundation_models import ClipMobile, NativeTextClip, OpenClip

    base = _mobile_like("custom_text")
    registry = {"MobileCLIP-S1": (lambda: base.model, 64, oc.HALF, oc.HALF, oc._tokenizer(1000, eot=999))}
    monkeypatch.setitem(sys.modules, "open_clip", oc.fake_open_clip_module(registry))
    fm = ClipMobile("s1", device=DEV)
    nat = fm.native(device_preprocess=False)
    assert isinstance(nat, NativeTextClip)
    with pytest.raises(TypeError):
        fm.native(image_tower="native")
    assert OpenClip.NATIVE_IMAGE_TOWER == "native" and ClipMobile.NATIVE_IMAGE_TOWER == "auto"
    tok = fm.tokenize(PROMPTS)
    with torch.no_grad():
        assert rel_err(nat.encode_text(tok), fm.model.encode_text(tok)) < 3e-5
