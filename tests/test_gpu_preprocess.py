"""GPU parity for K12 (image preprocessing on the device) through the C ABI: bit-exact against the Pillow golden
vectors and against the CPU oracle on ragged batches at realistic sizes."""
import numpy as np
import pytest
import torch

import oracle
from semanticlens_amd import _native as N
from semanticlens_amd.foundation_models import DevicePreprocess

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cases(g):
    for i in range(int(g["n"])):
        S, mode, interp = (int(v) for v in g[f"cfg{i}"])
        yield i, g[f"img{i}"], S, ["shortest", "squash"][mode], ["bicubic", "bilinear"][interp]


def _run(imgs, S, mean, std, mode="shortest", interp="bicubic"):
    pp = DevicePreprocess(S, mean, std, mode, interp, device=DEV)
    buf, plan, info = pp.pack(imgs)
    f, u8 = N.preprocess(buf.to(DEV), plan, info, S, pp.mean, pp.std, interp, want_u8=True)
    torch.cuda.synchronize()
    return f.cpu().numpy(), u8.cpu().numpy()


def test_golden_cases_bit_exact(golden):
    g = golden("preprocess")
    for i, img, S, mode, interp in _cases(g):
        f, u8 = _run([img], S, g["mean"], g["std"], mode, interp)
        if f"u8_{i}" in g:
            assert np.array_equal(u8[0], g[f"u8_{i}"]), f"case {i}: bytes differ from Pillow"
            assert np.array_equal(f[0], g[f"f32_{i}"]), f"case {i}: floats differ from torch"
        else:
            assert int(u8[0].astype(np.int64).sum()) == int(g[f"u8_{i}_sum"])
            assert np.array_equal(u8[0][::7, ::5], g[f"u8_{i}_sample"])
            assert np.array_equal(f[0][:, ::7, ::5], g[f"f32_{i}_sample"])


def test_ragged_batch_equals_per_image_results(golden):
    g = golden("preprocess")
    group = [(img, i) for i, img, S, mode, interp in _cases(g) if S == 32 and mode == "shortest" and interp == "bicubic"]
    f, u8 = _run([img for img, _ in group], 32, g["mean"], g["std"])
    for b, (_, i) in enumerate(group):
        assert np.array_equal(u8[b], g[f"u8_{i}"])
        assert np.array_equal(f[b], g[f"f32_{i}"])


@pytest.mark.parametrize("mode,interp", [("shortest", "bicubic"), ("squash", "bicubic"), ("shortest", "bilinear")])
def test_realistic_sizes_match_oracle(mode, interp):
    rng = np.random.default_rng(11)
    sizes = [(500, 375), (375, 500), (333, 500), (224, 224), (1200, 900), (256, 341), (97, 640), (480, 640)]
    imgs = []
    for h, w in sizes:
        yy, xx = np.mgrid[0:h, 0:w]
        base = 128 + 100 * np.sin(yy[..., None] / 9.0 + np.arange(3)) * np.cos(xx[..., None] / 13.0)
        imgs.append(np.clip(base + rng.normal(0, 50, (h, w, 3)), 0, 255).astype(np.uint8))
    mean, std = (0.5, 0.4, 0.3), (0.2, 0.25, 0.3)
    f, u8 = _run(imgs, 224, mean, std, mode, interp)
    for b, img in enumerate(imgs):
        eu8, ef = oracle.preprocess(img, 224, mean, std, mode, interp)
        assert np.array_equal(u8[b], eu8), f"image {b} {img.shape}"
        assert np.array_equal(f[b], ef), f"image {b} {img.shape}"


def test_identity_size_is_pure_normalisation():
    x = torch.randint(0, 256, (6, 64, 64, 3), dtype=torch.uint8)
    pp = DevicePreprocess(64, device=DEV)
    out = pp(x.to(DEV)).cpu()
    mean, std = torch.tensor(pp.mean), torch.tensor(pp.std)
    exp = x.permute(0, 3, 1, 2).float().div(255).sub(mean[None, :, None, None]).div(std[None, :, None, None])
    assert torch.equal(out, exp)
    one = pp(x[0].numpy())
    assert one.shape == (3, 64, 64) and torch.equal(one.cpu(), exp[0])


def test_native_clip_with_device_preprocess_matches_host_transform():
    """Lens-level: NativeClip(preprocess=DevicePreprocess) on raw uint8 images == the same tower fed the oracle's
    (Pillow-exact) host transform."""
    import synth
    from semanticlens_amd.foundation_models.native_clip import NativeClip

    S = 64
    base = synth.SyntheticClip(device=DEV, seed=0, embed_dim=64, image_size=S, patch=16, v_width=128, v_layers=2,
                               v_heads=2, t_width=128, t_layers=1, t_heads=2, vocab=49408)
    pp = DevicePreprocess(S, device=DEV)
    fm = NativeClip(base, preprocess=pp)
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(300, 260), (S, S), (250, 400)]]
    got = fm.encode_image(fm.preprocess(imgs)).cpu()
    host = np.stack([oracle.preprocess(i, S, pp.mean, pp.std)[1] for i in imgs])
    exp = fm.encode_image(torch.from_numpy(host).to(DEV)).cpu()
    assert torch.equal(got, exp)


def test_long_filters_and_large_sources_match_pillow_golden(golden):
    """1500x1000 -> 32 (127 taps), 37x2900 squash, 2000x3000 -> 224: procedurally generated sources, Pillow's output."""
    from test_oracle_golden import _procedural_image

    g = golden("preprocess")
    for j in range(int(g["n_proc"])):
        h, w, S, mode, interp = (int(v) for v in g[f"proc_cfg{j}"])
        f, u8 = _run([_procedural_image(h, w)], S, g["mean"], g["std"], ["shortest", "squash"][mode], ["bicubic", "bilinear"][interp])
        assert int(u8[0].astype(np.int64).sum()) == int(g[f"proc_u8_{j}_sum"]), j
        assert np.array_equal(u8[0][::3, ::3], g[f"proc_u8_{j}_sample"]) and np.array_equal(f[0][:, ::5, ::5], g[f"proc_f32_{j}_sample"]), j
