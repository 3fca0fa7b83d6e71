"""K11: the native transformer towers (HIP kernels through the C ABI) against the torch modules they were built
from — the same `AbstractVLM` seam the reference's OpenClip sits behind (foundation_models/clip.py:103-135).
open_clip itself is not installed, so the reference pins shapes only (tests/foundation_models/test_clip.py);
here parity is torch-module (fp32, same weights) vs native, tolerance 1e-4 of the feature scale."""
import numpy as np
import pytest
import torch

import synth
from semanticlens_amd import _native as N
from semanticlens_amd.foundation_models.native_clip import NativeClip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


def test_primitives_against_torch():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(300, 200, device=DEV, generator=g)
    w = torch.randn(130, 200, device=DEV, generator=g) * 0.1
    b = torch.randn(130, device=DEV, generator=g)
    r = torch.randn(300, 130, device=DEV, generator=g)
    ref = x @ w.T + b
    assert rel_err(N.linear(x, w, b), ref) < 1e-5
    assert rel_err(N.linear(x, w, b, act=N.SL_ACT_GELU), torch.nn.functional.gelu(ref)) < 1e-5
    assert rel_err(N.linear(x, w, b, act=N.SL_ACT_QUICKGELU), ref * torch.sigmoid(1.702 * ref)) < 1e-5
    out = r.clone()
    N.linear(x, w, b, residual=out, out=out)
    assert rel_err(out, ref + r) < 1e-5
    gam, bet = torch.randn(200, device=DEV, generator=g), torch.randn(200, device=DEV, generator=g)
    assert rel_err(N.layernorm(x, gam, bet, 1e-5), torch.nn.functional.layer_norm(x, (200,), gam, bet, 1e-5)) < 1e-5
    # attention vs torch.nn.MultiheadAttention (packed in_proj), plain and causal, T below and above one wave
    for T, causal in ((50, False), (77, True), (130, True), (7, False)):
        B, H = 3, 2
        mha = torch.nn.MultiheadAttention(H * 64, H, batch_first=True).to(DEV)
        xx = torch.randn(B, T, H * 64, device=DEV, generator=g)
        mask = torch.full((T, T), float("-inf"), device=DEV).triu_(1) if causal else None
        want = mha(xx, xx, xx, need_weights=False, attn_mask=mask)[0]
        qkv = N.linear(xx.reshape(B * T, -1), mha.in_proj_weight.detach(), mha.in_proj_bias.detach())
        att = N.attention(qkv, B, T, H, 64, causal)
        got = N.linear(att, mha.out_proj.weight.detach(), mha.out_proj.bias.detach()).reshape(B, T, -1)
        assert rel_err(got, want) < 1e-5, (T, causal)
    img = torch.randn(2, 3, 32, 64, device=DEV, generator=g)
    conv = torch.nn.Conv2d(3, 24, 16, 16, bias=False).to(DEV)
    want = conv(img).flatten(2).transpose(1, 2).reshape(-1, 24)
    assert rel_err(N.linear(N.patchify(img, 16), conv.weight.detach().reshape(24, -1)), want) < 1e-5
    # split-bf16 operands: hi + lo reconstructs the value to ~2^-17, and the bf16x3 GEMM matches fp32 results
    sp = N.Split.of(x)
    assert ((sp.hi.float() + sp.lo.float()) - x).abs().max().item() <= 2.0 ** -16 * x.abs().max().item()
    # layout: rows of 2 Kp values, each 32-wide k-tile = one 128-byte line [hi32 | lo32], zero padding up to Kp = 224
    assert sp.buf.shape == (300, 2 * 224) and sp.buf.view(300, 7, 2, 32)[:, 6, :, 8:].float().abs().max().item() == 0.0
    hi_ref = x.to(torch.bfloat16)
    assert torch.equal(sp.hi, hi_ref) and torch.equal(sp.lo, (x - hi_ref.float()).to(torch.bfloat16))
    ws = N.Split.of(w)
    # any K works now (padding is part of the format): K = 100 is not even a multiple of 8
    xa, wa = torch.randn(70, 100, device=DEV, generator=g), torch.randn(33, 100, device=DEV, generator=g)
    assert rel_err(N.linear3(N.Split.of(xa), N.Split.of(wa)), xa @ wa.T) < 1e-5
    assert rel_err(N.linear3(sp, ws, b), ref) < 1e-5
    assert rel_err(N.linear3(sp, ws, b, act=N.SL_ACT_GELU), torch.nn.functional.gelu(ref)) < 1e-5
    out = r.clone()
    N.linear3(sp, ws, b, residual=out, out=out)
    assert rel_err(out, ref + r) < 1e-5
    o2 = N.linear3(sp, ws, b, out_split=N.Split(300, 130, DEV))
    assert rel_err(o2.hi.float() + o2.lo.float(), ref) < 1e-5
    l2 = N.layernorm(x, gam, bet, 1e-5, out_split=N.Split(300, 200, DEV))
    assert rel_err(l2.hi.float() + l2.lo.float(), torch.nn.functional.layer_norm(x, (200,), gam, bet, 1e-5)) < 1e-5


@pytest.mark.parametrize("arch", [
    dict(embed_dim=64, image_size=64, patch=16, v_width=128, v_layers=2, v_heads=2, ctx=16, vocab=49408, t_width=128, t_layers=2, t_heads=2),
    dict(embed_dim=48, image_size=64, patch=16, v_width=160, v_layers=2, v_heads=2, ctx=16, vocab=49408, t_width=64, t_layers=1, t_heads=2),  # head_dim 80 / 32
    dict(),  # CLIP ViT-B/32: 12 x 768 image tower (50 tokens), 12 x 512 text tower (77 tokens), 512-d joint space
])
@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
def test_native_towers_match_torch_modules(arch, gemm):
    fm = synth.SyntheticClip(device=DEV, seed=3, **arch)
    nat = NativeClip(fm, gemm=gemm)
    size = arch.get("image_size", 224)
    u8 = synth.synth_images_u8(torch.arange(6, device=DEV), size=size)
    x = fm.preprocess(u8)
    want = fm.encode_image(x)
    got = nat.encode_image(x)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert rel_err(got, want) < 1e-4, rel_err(got, want)
    toks = fm.tokenize(["a photo of a cat", "dog", "a very long prompt with many many words in it " * 3])
    want_t = fm.encode_text(toks)
    got_t = nat.encode_text(toks)
    assert rel_err(got_t, want_t) < 1e-4, rel_err(got_t, want_t)
    # dropping the padding after the batch's last end-of-text token (causal tower) changes no pooled bit
    short = fm.tokenize(["a photo of a cat", "dog", "two red wheels"])
    nat.text.truncate = False
    full = nat.encode_text(short)
    nat.text.truncate = True
    assert torch.equal(nat.encode_text(short), full)
    # last block computed for the pooled rows only: same bits as running it over every token
    nat.text.pool_shortcut = nat.vision.pool_shortcut = False
    full_t, full_v = nat.encode_text(toks), nat.encode_image(x)
    nat.text.pool_shortcut = nat.vision.pool_shortcut = True
    assert torch.equal(nat.encode_text(toks), full_t) and torch.equal(nat.encode_image(x), full_v)
    # cosine between the two implementations' features ~ 1
    cos = torch.nn.functional.cosine_similarity(got, want, dim=-1)
    assert (1 - cos).abs().max().item() < 1e-6


def test_native_clip_drops_into_lens_pipeline():
    """Same concept_db (within 1e-4 of the embedding scale) whether the AbstractVLM is the torch model or NativeClip."""
    from semanticlens_amd import Lens
    from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators

    arch = dict(embed_dim=64, image_size=64, patch=16, v_width=128, v_layers=2, v_heads=2, ctx=16, vocab=49408, t_width=128, t_layers=2, t_heads=2)
    fm = synth.SyntheticClip(device=DEV, seed=5, **arch)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.ReLU()).to(DEV).eval()

    def build():
        ds_m = synth.SyntheticImageDataset(24, "model", size=64)
        ds_f = synth.SyntheticImageDataset(24, "fm", size=64)
        return ActivationComponentVisualizer(model, ds_m, ds_f, ["1"], num_samples=4, aggregate_fn=aggregators.aggregate_conv_max, tie_mode="aten")

    db_torch = Lens(fm, device=DEV).compute_concept_db(build(), batch_size=8)
    db_native = Lens(NativeClip(fm), device=DEV).compute_concept_db(build(), batch_size=8)
    assert rel_err(db_native["1"], db_torch["1"]) < 1e-4
    probe_t = Lens(fm, device=DEV).text_probing(["cat", "dog"], {"1": db_torch["1"].mean(1)})
    probe_n = Lens(NativeClip(fm), device=DEV).text_probing(["cat", "dog"], {"1": db_native["1"].mean(1)})
    np.testing.assert_allclose(probe_n["1"].cpu().numpy(), probe_t["1"].cpu().numpy(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("T,causal,D", [(1, False, 64), (31, True, 64), (32, False, 64), (33, True, 64), (50, False, 64), (64, True, 64),
                                         (77, True, 64), (130, False, 64), (197, False, 64), (256, True, 64), (257, True, 64), (300, False, 64),
                                         (577, True, 64), (600, False, 64), (50, False, 32), (77, True, 72), (257, False, 80), (40, True, 88),
                                         (200, True, 96), (65, False, 104), (300, True, 128)])
def test_attention_matches_fp32_softmax_reference(T, causal, D):
    """sl_attention (fp32 matrix-core kernel) against softmax(q k^T / sqrt(D)) v in torch fp32, head by head, for every
    built head_dim and sequence lengths on both sides of the 32-row tiles and of the LDS chunk; the split (hi, lo)
    output form carries the same values."""
    g = torch.Generator(device=DEV).manual_seed(T + D)
    B, H = 2, 3
    qkv = torch.randn(B * T, 3 * H * D, device=DEV, generator=g)
    q, k, v = (qkv.reshape(B, T, 3, H, D)[:, :, i].permute(0, 2, 1, 3) for i in range(3))  # (B, H, T, D)
    s = (q @ k.transpose(-1, -2)) * (D ** -0.5)
    if causal:
        s = s + torch.full((T, T), float("-inf"), device=DEV).triu_(1)
    want = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, H * D)
    got = N.attention(qkv, B, T, H, D, causal)
    assert (got - want).abs().max().item() < 2e-6 * max(1.0, want.abs().max().item()), (T, causal, D)
    sp = N.Split(B * T, H * D, DEV)
    N.attention(qkv, B, T, H, D, causal, out_split=sp)
    assert (sp.hi.float() + sp.lo.float() - got).abs().max().item() <= 2.0 ** -16 * got.abs().max().item() + 1e-9
    with pytest.raises(ValueError):
        N.attention(qkv, B, T, H, 48, causal)
    # the split-bf16 x3 form (sl_attention_bf16x3: what the towers run between two bf16x3 GEMMs): both products as
    # a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16 matrix cores — the dropped lo*lo terms are 2^-18 per product
    got3 = N.attention(qkv, B, T, H, D, causal, bf16x3=True)
    assert (got3 - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()), (T, causal, D, (got3 - want).abs().max().item())
    sp3 = N.Split(B * T, H * D, DEV)
    N.attention(qkv, B, T, H, D, causal, out_split=sp3, bf16x3=True)
    assert (sp3.hi.float() + sp3.lo.float() - got3).abs().max().item() <= 2.0 ** -16 * got3.abs().max().item() + 1e-9
    with pytest.raises(ValueError):
        N.attention(qkv, B, T, H, 48, causal, bf16x3=True)


@pytest.mark.parametrize("cols", [4, 130, 200, 768, 1024, 1028, 2048])
def test_layernorm_paths(cols):
    """register-cached fast path (cols % 4 == 0, <= 1024) and the three-pass fallback, fp32 and split outputs, and a
    strided input (every 5th row, as the class-token pick of the vision tower does)."""
    g = torch.Generator(device=DEV).manual_seed(cols)
    x = torch.randn(77, cols, device=DEV, generator=g) * 3 + 1
    gam, bet = torch.randn(cols, device=DEV, generator=g), torch.randn(cols, device=DEV, generator=g)
    want = torch.nn.functional.layer_norm(x, (cols,), gam, bet, 1e-5)
    assert rel_err(N.layernorm(x, gam, bet, 1e-5), want) < 1e-5
    sp = N.layernorm(x, gam, bet, 1e-5, out_split=N.Split(77, cols, DEV))
    assert rel_err(sp.hi.float() + sp.lo.float(), want) < 1e-5
    assert rel_err(N.layernorm(x, gam, bet, 1e-5, rows=16, x_row_stride=5 * cols), want[::5]) < 1e-5


def test_linear3_random_shapes_against_fp64():
    """Packed split operands + bf16x3 GEMM on ragged shapes (K not a multiple of 32 / 8, single rows, tile edges),
    with bias, against an fp64 matmul; also through the split-output form."""
    rng = np.random.default_rng(123)
    g = torch.Generator(device=DEV).manual_seed(123)
    shapes = [(1, 1, 1), (1, 7, 33), (129, 127, 31), (255, 257, 95), (300, 130, 200), (64, 512, 1152), (513, 260, 72)]
    shapes += [tuple(int(v) for v in rng.integers(1, 400, 3)) for _ in range(12)]
    for M, Nn, K in shapes:
        x = torch.randn(M, K, device=DEV, generator=g)
        w = torch.randn(Nn, K, device=DEV, generator=g)
        b = torch.randn(Nn, device=DEV, generator=g)
        ref = (x.double() @ w.double().T + b.double())
        # three-product split: ~2^-16 relative per product, random signs -> ~2^-16 sqrt(K) absolute for unit-variance data
        tol = 8 * 2.0 ** -16 * K ** 0.5 + 1e-6
        got = N.linear3(N.Split.of(x), N.Split.of(w), b)
        assert (got.double() - ref).abs().max().item() < tol, (M, Nn, K)
        sp = N.linear3(N.Split.of(x), N.Split.of(w), b, out_split=N.Split(M, Nn, DEV))
        assert ((sp.hi.float() + sp.lo.float()).double() - ref).abs().max().item() < tol + 2.0 ** -15 * ref.abs().max().item(), (M, Nn, K)
        if sp.kp != Nn:  # the padding of a split output stays zero (the next GEMM reads it)
            assert sp.buf.view(M, sp.kp // 32, 2, 32)[:, -1, :, Nn % 32:].float().abs().max().item() == 0.0


# ---- SigLIP-layout towers (clip.py:190-211 SigLipV2; BASELINE configs[3]: SigLIP-so400m) ------------------------------
def test_gelu_tanh_and_attention_pool_primitives():
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(200, 96, device=DEV, generator=g)
    w = torch.randn(50, 96, device=DEV, generator=g) * 0.2
    b = torch.randn(50, device=DEV, generator=g)
    ref = torch.nn.functional.gelu(x @ w.T + b, approximate="tanh")
    assert rel_err(N.linear(x, w, b, act=N.SL_ACT_GELU_TANH), ref) < 1e-5
    assert rel_err(N.linear3(N.Split.of(x), N.Split.of(w), b, act=N.SL_ACT_GELU_TANH), ref) < 1e-5
    # one query per head over T keys == torch MultiheadAttention with a single query token
    for B, T, H, hd in ((3, 16, 2, 72), (2, 197, 4, 64), (1, 729, 16, 72), (5, 3, 1, 128)):
        W = H * hd
        mha = torch.nn.MultiheadAttention(W, H, batch_first=True).to(DEV)
        probe = torch.randn(1, 1, W, device=DEV, generator=g)
        xx = torch.randn(B, T, W, device=DEV, generator=g)
        want = mha(probe.expand(B, 1, W), xx, xx, need_weights=False)[0][:, 0]
        wq, wkv = mha.in_proj_weight.detach()[:W], mha.in_proj_weight.detach()[W:]
        bq, bkv = mha.in_proj_bias.detach()[:W], mha.in_proj_bias.detach()[W:]
        q = N.linear(probe.reshape(1, W), wq, bq).reshape(W)
        kv = N.linear(xx.reshape(B * T, W), wkv, bkv)
        got = N.linear(N.attention_pool(q, kv, B, T, H, hd), mha.out_proj.weight.detach(), mha.out_proj.bias.detach())
        assert rel_err(got, want) < 1e-5, (B, T, H, hd)


@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
@pytest.mark.parametrize("geom", [dict(width=144, layers=2, heads=2, mlp=288, image_size=64, patch=16, ctx=16, vocab=1000),
                                  dict(width=1152, layers=2, heads=16, mlp=4304, image_size=224, patch=14, ctx=64, vocab=32000)])
def test_native_siglip_matches_the_torch_module(geom, gemm):
    """MAP-pooled image tower without class token, non-causal text tower pooled at the last position, head_dim 72
    (second case: the SigLIP-so400m geometry at two layers) against transformers' SiglipModel with the same weights."""
    from semanticlens_amd.foundation_models.native_clip import NativeSigLip

    base = synth.SyntheticSigLip(device=DEV, seed=3, **geom)
    fm = NativeSigLip(base, gemm=gemm)
    g = torch.Generator(device=DEV).manual_seed(1)
    img = torch.randn(5, 3, geom["image_size"], geom["image_size"], device=DEV, generator=g)
    want_i = base.encode_image(img)
    got_i = fm.encode_image(img)
    assert got_i.shape == want_i.shape == (5, geom["width"])
    assert rel_err(got_i, want_i) < 2e-5, rel_err(got_i, want_i)
    tok = base.tokenize(["a photo of a cat", "zebra", "a very long prompt about " + "stripes " * 80, ""])
    want_t = base.encode_text(tok)
    got_t = fm.encode_text(tok)
    assert got_t.shape == want_t.shape == (4, geom["width"])
    assert rel_err(got_t, want_t) < 2e-5, rel_err(got_t, want_t)


def test_config3_end_to_end_vit_probe_siglip_embed_text_probing():
    """BASELINE configs[3] at reduced depth: ViT-B/16-geometry blocks probed with the token-max aggregator, SigLIP-so400m
    geometry (two layers) as the foundation model running natively, `text_probing` over 10 000 prompts through the native
    text tower and the cosine GEMM — against the same pipeline with the torch SigLIP module."""
    from semanticlens_amd import Lens
    from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators
    from semanticlens_amd.foundation_models.native_clip import NativeSigLip

    torch.manual_seed(0)

    class Vit(torch.nn.Module):  # patch embedding + 2 encoder blocks of width 768 over 197 tokens
        def __init__(self):
            super().__init__()
            self.patch = torch.nn.Conv2d(3, 768, 16, 16)
            self.cls = torch.nn.Parameter(torch.randn(1, 1, 768) * 0.02)
            self.blocks = torch.nn.ModuleList([synth._Block(768, 12) for _ in range(2)])
            self.name = "vit-b16-2blocks"

        def forward(self, x):
            t = torch.cat([self.cls.expand(x.shape[0], 1, 768), self.patch(x).flatten(2).transpose(1, 2)], 1)
            for b in self.blocks:
                t = b(t)
            return t[:, 0]

    n, k = 96, 8
    u8 = synth.synth_images_u8(torch.arange(n, device=DEV)).cpu()

    class DS(torch.utils.data.Dataset):
        def __init__(self, mode):
            self.mode, self.name = mode, f"c3-{n}"

        def __len__(self):
            return n

        def __getitem__(self, i):
            if self.mode == "model":
                return synth.normalize_u8(u8[i:i + 1], synth.IMAGENET_MEAN, synth.IMAGENET_STD)[0], 0
            return u8[i]

    base = synth.SyntheticSigLip(device=DEV, seed=5, width=1152, layers=2, heads=16, mlp=4304, image_size=224, patch=14, ctx=64)
    model = Vit().to(DEV).eval()
    dbs, probes = {}, {}
    words = ["zebra", "stripe", "wheel", "sky", "grass", "dog", "cat", "red", "round", "metal"]
    prompts = [f"a photo of a {words[i % 10]} {words[(i // 10) % 10]} {i}" for i in range(10000)]
    for tag, fm in (("native", NativeSigLip(base)), ("torch", base)):
        cv = ActivationComponentVisualizer(model, DS("model"), DS("fm"), ["blocks.0", "blocks.1"], num_samples=k,
                                           aggregate_fn=aggregators.aggregate_transformer_max, tie_mode="aten")
        lens = Lens(fm, device=DEV)
        dbs[tag] = lens.compute_concept_db(cv, batch_size=32)
        agg = {name: v.mean(1) for name, v in dbs[tag].items()}
        probes[tag] = lens.text_probing(prompts, agg, batch_size=2500)
    for name in ("blocks.0", "blocks.1"):
        assert dbs["native"][name].shape == (768, k, 1152)
        scale = dbs["torch"][name].abs().max().item()
        assert (dbs["native"][name] - dbs["torch"][name]).abs().max().item() < 1e-4 * scale
        assert probes["native"][name].shape == (10000, 768)
        assert (probes["native"][name] - probes["torch"][name]).abs().max().item() < 1e-4  # north_star: cosines within 1e-4


def test_half_batches_on_two_streams_keep_every_bit(monkeypatch):
    """The image tower can cut a large batch in SL_ENC_STREAMS chunks, one HIP stream each.  Samples are independent through
    the tower — GEMM rows, attention per image — and every kernel rounds an element the same way wherever its row sits in a
    tile (the GELU epilogue did not: fma contraction differed between unrolled copies), so: same bits either way, also at a
    chunk boundary (96 x 50 rows) that is not a multiple of the 256-row tile."""
    from semanticlens_amd.foundation_models import native_clip as nc

    fm = synth.SyntheticClip(device=DEV, seed=3)
    nat = NativeClip(fm)
    x = fm.preprocess(synth.synth_images_u8(torch.arange(192, device=DEV)))
    monkeypatch.setenv("SL_ENC_STREAMS", "1")
    one = nat.encode_image(x)
    monkeypatch.setenv("SL_ENC_STREAMS", "2")
    assert 96 * nat.vision.chunk_rows(x) >= 4096  # the split is taken
    two = nat.encode_image(x)
    assert torch.equal(one, two)
    monkeypatch.setenv("SL_ENC_STREAMS", "3")
    three = nc._in_chunks(nat.vision._encode, x, nat.vision.chunk_rows(x), min_rows=0)
    assert torch.equal(one, three)
    assert torch.equal(nat.encode_image(x[:5]), one[:5])  # small batches: one call on the current stream


# ---- the comparison object is FLOAT64, the bar is ABSOLUTE ---------------------------------------------------
# north_star: "cosine/embedding values within 1e-4 fp32".  The towers above are checked against fp32 torch modules with a
# relative bar (a smoke bar: both sides carry fp32 rounding); here the same weights run in float64 on the device and the
# native features must sit within 1e-4 ABSOLUTE of them un-normalised, and the image-text cosines within 1e-5.
def _fp64_features(fm, x, toks):
    import copy

    ref = copy.deepcopy(fm.model).double()
    with torch.no_grad():
        if hasattr(ref, "vision_model"):  # transformers' SiglipModel
            return ref.vision_model(pixel_values=x.double()).pooler_output, ref.text_model(input_ids=toks).pooler_output
        return ref.encode_image(x.double()), ref.encode_text(toks)


def _assert_within_fp64(got_i, got_t, want_i, want_t, tag):
    d_i = (got_i.double() - want_i).abs().max().item()
    d_t = (got_t.double() - want_t).abs().max().item()
    assert d_i < 1e-4 and d_t < 1e-4, (tag, d_i, d_t, want_i.abs().max().item(), want_t.abs().max().item())
    cos = torch.nn.functional.normalize(got_i.double(), dim=-1) @ torch.nn.functional.normalize(got_t.double(), dim=-1).T
    cos64 = torch.nn.functional.normalize(want_i, dim=-1) @ torch.nn.functional.normalize(want_t, dim=-1).T
    d_c = (cos - cos64).abs().max().item()
    assert d_c < 1e-5, (tag, d_c)
    return d_i, d_t, d_c


@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
def test_vit_b32_towers_within_1e4_absolute_of_float64(gemm):
    """CLIP ViT-B/32 geometry (12 x 768 image tower, 12 x 512 text tower, 512-d joint space), both GEMM modes."""
    fm = synth.SyntheticClip(device=DEV, seed=3)
    x = fm.preprocess(synth.synth_images_u8(torch.arange(16, device=DEV)))
    toks = fm.tokenize(["a photo of a cat", "dog", "a very long prompt with many many words in it " * 3, "two red wheels on wet grass",
                        "sky", "x", "a striped zebra near the river bank at dawn", "metal text on a wooden face"])
    want_i, want_t = _fp64_features(fm, x, toks)
    nat = NativeClip(fm, gemm=gemm)
    _assert_within_fp64(nat.encode_image(x), nat.encode_text(toks), want_i, want_t, gemm)
    # the torch fp32 module itself sits at the same distance from float64 (what "fp32-class accuracy" means)
    d_torch = (fm.encode_image(x).double() - want_i).abs().max().item()
    assert d_torch < 1e-4


@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
def test_so400m_towers_within_1e4_absolute_of_float64(gemm):
    """SigLIP-so400m geometry at full depth (27 x 1152, 16 heads of 72, MLP 4304, patch 14 -> 256 tokens, ctx 64):
    BASELINE configs[3]'s embed model, transformers' `SiglipModel` with random weights in float64 as the reference."""
    from semanticlens_amd.foundation_models import NativeSigLip

    fm = synth.SyntheticSigLip(device=DEV)
    x = fm.preprocess(synth.synth_images_u8(torch.arange(4, device=DEV)))
    toks = fm.tokenize(["a photo of a cat", "dog", "a striped zebra near the river bank at dawn", "metal text on a wooden face"])
    want_i, want_t = _fp64_features(fm, x, toks)
    nat = NativeSigLip(fm, gemm=gemm)
    _assert_within_fp64(nat.encode_image(x), nat.encode_text(toks), want_i, want_t, gemm)


# ---- massive activations ------------------------------------------------------------------------------------------------
# Pretrained CLIP / SigLIP checkpoints (what `OpenClip(url)` loads, foundation_models/clip.py:52-62) carry "massive activations":
# a few residual channels sit at 100-300 on every token from an early block on, and the class token has a large norm.  Every
# tower test above is random-init (residual stream of order 1), where a 3-product bf16 split (per-product relative error
# ~2^-17) is trivially inside 1e-4 absolute.  Here the same towers are given such channels (a bias on two residual channels
# written by early MLPs, a 60x class embedding) and BOTH GEMM modes must still sit within north_star's 1e-4 ABSOLUTE of the
# same weights in float64.
def _inject_massive_activations(fm):
    m = fm.model
    with torch.no_grad():
        if hasattr(m, "vision_model"):  # transformers' SiglipModel
            enc_v, enc_t = m.vision_model.encoder.layers, m.text_model.encoder.layers
            enc_v[1].mlp.fc2.bias[5] += 200.0
            enc_v[3].mlp.fc2.bias[300] -= 300.0
            enc_t[1].mlp.fc2.bias[7] += 150.0
            return [enc_v[4], enc_t[2]]
        m.visual.blocks[1].mlp[2].bias[5] += 150.0
        m.visual.blocks[3].mlp[2].bias[300] -= 250.0
        m.class_embedding.mul_(60.0)
        m.text.blocks[1].mlp[2].bias[7] += 120.0
        return [m.visual.blocks[4], m.text.blocks[2]]


def _residual_maxima(fm, probes, x, toks):
    seen = []
    hooks = [p.register_forward_hook(lambda mod, i, o: seen.append((o[0] if isinstance(o, tuple) else o).abs().max().item())) for p in probes]
    fm.encode_image(x), fm.encode_text(toks)
    for h in hooks:
        h.remove()
    return seen


@pytest.mark.parametrize("arch", ["vit_b32", "so400m"])
def test_towers_with_massive_activations_within_1e4_absolute_of_float64(arch):
    from semanticlens_amd.foundation_models import NativeSigLip

    if arch == "vit_b32":
        fm, cls, n_img = synth.SyntheticClip(device=DEV, seed=3), NativeClip, 8
    else:
        fm, cls, n_img = synth.SyntheticSigLip(device=DEV), NativeSigLip, 4
    probes = _inject_massive_activations(fm)
    x = fm.preprocess(synth.synth_images_u8(torch.arange(n_img, device=DEV)))
    toks = fm.tokenize(["a photo of a cat", "dog", "a striped zebra near the river bank at dawn", "metal text on a wooden face"])
    maxima = _residual_maxima(fm, probes, x, toks)
    assert min(maxima) > 100.0, maxima  # the residual streams of both towers really carry the massive channels
    want_i, want_t = _fp64_features(fm, x, toks)
    d_torch = max((fm.encode_image(x).double() - want_i).abs().max().item(), (fm.encode_text(toks).double() - want_t).abs().max().item())
    for gemm in ("bf16x3", "f32"):
        nat = cls(fm, gemm=gemm)
        d_i, d_t, d_c = _assert_within_fp64(nat.encode_image(x), nat.encode_text(toks), want_i, want_t, (arch, gemm))
        print(f"massive activations [{arch} {gemm}]: residual max {max(maxima):.0f}; |d image| {d_i:.2e} |d text| {d_t:.2e} |d cos| {d_c:.2e}; "
              f"torch fp32 {d_torch:.2e}; feature scale {want_i.abs().max().item():.2f} / {want_t.abs().max().item():.2f}")


# ---- CLIP-ResNet (open_clip ModifiedResNet: `OpenClip("RN50", ...)`, BASELINE configs[0]'s embed model) ------------------------
def test_tokens_from_map_and_per_image_query_pool_primitives():
    g = torch.Generator(device=DEV).manual_seed(11)
    for B, C, S in ((3, 2048, 49), (2, 100, 7), (1, 64, 196), (5, 192, 1)):
        fmap = torch.randn(B, C, S, device=DEV, generator=g)
        pos = torch.randn(S + 1, C, device=DEV, generator=g)
        got = N.tokens_from_map(fmap.reshape(B, C, S, 1), pos)
        x = fmap.permute(0, 2, 1)
        want = torch.cat([x.mean(1, keepdim=True), x], 1) + pos
        assert got.shape == (B, S + 1, C) and torch.allclose(got, want, rtol=0, atol=2e-6), (B, C, S)
        assert torch.equal(got[:, 1:], want[:, 1:])  # the plain tokens are one add: exact
    B, T, H, hd = 4, 50, 32, 64
    W = H * hd
    kv = torch.randn(B * T, 2 * W, device=DEV, generator=g)
    q = torch.randn(B, W, device=DEV, generator=g)
    got = N.attention_pool_q(q, kv, B, T, H, hd)
    k = kv[:, :W].reshape(B, T, H, hd).permute(0, 2, 1, 3).double()
    v = kv[:, W:].reshape(B, T, H, hd).permute(0, 2, 1, 3).double()
    p = torch.softmax((q.reshape(B, H, 1, hd).double() @ k.transpose(-1, -2)) / hd**0.5, -1)
    want = (p @ v).reshape(B, W)
    assert (got.double() - want).abs().max().item() < 1e-5
    # stride 0 = one probe for every image (what sl_attention_pool has always done)
    assert torch.equal(N.attention_pool(q[0].contiguous(), kv, B, T, H, hd), N.attention_pool_q(q[:1].expand(B, W), kv, B, T, H, hd))


def _seeded_trunk_map(scale: float, seed: int = 5):
    """A deterministic `(8, 2048, 7, 7)` post-ReLU map generated on the HOST (no MIOpen output, no device generator) with the
    statistics a random-init ModifiedResNet trunk hands its attention pool: every channel has its own positive level shared by
    all positions and images (so the mean token — the pool's query — is large and the softmax logits grow with `scale`**2),
    plus per-position variation; `scale` = the magnitude of its largest entries (round 4 measured 22.5 on the driver's box)."""
    g = torch.Generator().manual_seed(seed)
    level = torch.rand(1, 2048, 1, 1, generator=g) ** 2  # most channels low, a few high
    m = (level * (1.0 + 0.5 * torch.randn(8, 2048, 7, 7, generator=g))).relu_()
    return (m * (scale / m.max())).to(DEV)


@pytest.mark.parametrize("gemm", ["bf16x3", "f32"])
def test_clip_rn50_attention_pool_head_on_the_kernels(gemm):
    """RN50-CLIP: the conv trunk on PyTorch, attention pool + `c_proj` + the whole text tower on the kernels.  The head is
    checked on SEEDED maps against the same weights in float64 with absolute bars (no convolution output in the comparison)."""
    import copy

    from semanticlens_amd.foundation_models.native_clip import NativeResNetVision

    fm = synth.SyntheticClipRN50(device=DEV, seed=4)
    nat = NativeClip(fm, gemm=gemm)
    assert isinstance(nat.vision, NativeResNetVision) and nat.vision.heads == 32 and nat.vision.head_dim == 64
    pool = fm.model.visual.attnpool
    assert pool is not None and not isinstance(pool, torch.nn.Identity)
    pool64 = copy.deepcopy(pool).double()
    # (i) unit-scale map: softmax logits of order 1-10 — north_star's 1e-4 ABSOLUTE on the un-normalised features
    # (ii) trunk-scale map (entries up to 20, logits in the hundreds): every fp32 implementation, torch's included, sits
    #      ~1e-4 from float64 here.  Bars: 5e-4 absolute on features of scale ~20 (2.5e-5 of the scale) AND at most twice
    #      the distance of torch's own fp32 attention pool on the same map.
    # (iii) scale 120: features of scale ~20 as on the round-4 driver box; no absolute bar is meaningful there (torch's own fp32
    #      pool is ~1e-4 from float64), only the one relative to torch
    for scale, bar in ((1.0, 1e-4), (20.0, 5e-4), (120.0, None)):
        fmap = _seeded_trunk_map(scale)
        got = nat.vision.head(fmap)
        assert torch.equal(got, nat.vision.head(fmap)), "the head is not run-to-run deterministic"
        with torch.no_grad():
            want = pool64(fmap.double())
            d_torch = (pool(fmap).double() - want).abs().max().item()
        d = (got.double() - want).abs().max().item()
        print(f"rn50 head [{gemm}] scale {scale}: native {d:.3e}  torch-fp32 {d_torch:.3e}  |features| {want.abs().max().item():.2f}")
        assert bar is None or d < bar, (gemm, scale, d, d_torch)
        assert d < max(2 * d_torch, 2e-5 if bar is None else 0.2 * bar), (gemm, scale, d, d_torch)
    # end to end behind the real trunk: `trunk()` must be the model's forward minus the pool.  MIOpen's convolutions are not
    # run-to-run deterministic at this batch size, so two calls of the trunk may differ in their last bits (amplified by the
    # pool's large logits): a 1e-3 relative smoke bar here, the arithmetic bars are the seeded ones above
    x = fm.preprocess(synth.synth_images_u8(torch.arange(8, device=DEV)))
    toks = fm.tokenize(["a photo of a cat", "dog", "two red wheels on wet grass", "sky"])
    got_i, got_t = nat.encode_image(x), nat.encode_text(toks)
    assert got_i.shape == (8, 1024) and got_t.shape == (4, 1024)
    assert rel_err(got_i, fm.encode_image(x)) < 1e-3
    fmap = nat.vision.trunk(x)
    assert fmap.shape == (8, 2048, 7, 7)
    assert rel_err(nat.vision.head(fmap), pool(fmap)) < 1e-4  # ONE trunk output through both pools
    want_i, want_t = _fp64_features(fm, x, toks)
    assert (got_t.double() - want_t).abs().max().item() < 1e-4
    cos = torch.nn.functional.normalize(got_i.double(), dim=-1) @ torch.nn.functional.normalize(got_t.double(), dim=-1).T
    cos64 = torch.nn.functional.normalize(want_i, dim=-1) @ torch.nn.functional.normalize(want_t, dim=-1).T
    assert (cos - cos64).abs().max().item() < 1e-4  # the fp32 conv trunk (38 M parameters, 53 layers) is inside this one
