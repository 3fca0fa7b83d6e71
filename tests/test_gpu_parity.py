"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bars (BASELINE.json north_star): top-k values and indices bit-exact; cosine / embedding values
within 1e-4 fp32 (we assert 1e-5).  Mean-type aggregators are order sensitive in fp32 and are
checked to 1 bf16 ulp on the stored value.
"""
import numpy as np
import pytest
import torch

import oracle
from semanticlens_amd import _native as N
from semanticlens_amd import scores
from semanticlens_amd.component_visualization import aggregators as agg
from semanticlens_amd.component_visualization.activation_caching import ActMax, ActMaxCache

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def feq(a, b):
    """float equality that treats NaN == NaN and -0 == +0 (torch.equal semantics + NaN)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=7e37), np.nan_to_num(b, nan=7e37))


def _parse(case):
    tag, k, B, every = case.split("|")
    return tag, int(k[2:]), int(B[2:]), bool(int(every[6:]))


# ---------------------------------------------------------------------------------------------- K3
def test_known_answer_of_reference_test(golden):
    g = golden("actmax_known_answer")
    for mode in ("aten", "total"):
        am = ActMax(n_collect=5, n_latents=3, tie_mode=mode)
        am.update(torch.from_numpy(g["acts1"]), torch.from_numpy(g["ids1"]))
        am.update(torch.from_numpy(g["acts2"]), torch.from_numpy(g["ids2"]))
        assert torch.allclose(am.activations[0], torch.tensor([0.9, 0.8, 0.2, 0.1, 0.0]).to(torch.bfloat16))
        assert torch.allclose(am.sample_ids[0], torch.tensor([2, 3, 1, 0, -1]))
        assert np.array_equal(bits(am.activations), g["vals"])
        assert np.array_equal(am.sample_ids.numpy(), g["ids"])


def test_golden_streams_aten_mode_bit_exact(golden):
    """tie_mode='aten' equals the reference after EVERY batch, ties included."""
    g = golden("actmax_streams")
    for idx, case in enumerate(g["cases"]):
        tag, k, B, every = _parse(str(case))
        acts = g[f"acts_{idx}"]
        am = ActMax(n_collect=k, n_latents=acts.shape[1], tie_mode="aten")
        step = 0
        for s in range(0, acts.shape[0], B):
            e = min(acts.shape[0], s + B)
            am.update(torch.from_numpy(acts[s:e]), torch.arange(s, e))
            if every:
                assert np.array_equal(bits(am.activations), g[f"vals_{idx}"][step]), (tag, k, B, step)
                assert np.array_equal(am.sample_ids.numpy(), g[f"ids_{idx}"][step]), (tag, k, B, step)
            step += 1
        assert np.array_equal(bits(am.activations), g[f"vals_{idx}"][-1]), (tag, k, B)
        assert np.array_equal(am.sample_ids.numpy(), g[f"ids_{idx}"][-1]), (tag, k, B)


def test_golden_streams_total_mode(golden):
    """tie_mode='total' == oracle(total) bit-exact; == the reference itself on tie-free inputs."""
    g = golden("actmax_streams")
    for idx, case in enumerate(g["cases"]):
        tag, k, B, every = _parse(str(case))
        acts = g[f"acts_{idx}"]
        N_, C = acts.shape
        am = ActMax(n_collect=k, n_latents=C, tie_mode="total")
        ref = oracle.ActMaxOracle(k, C, oracle.MODE_TOTAL)
        for s in range(0, N_, B):
            e = min(N_, s + B)
            am.update(torch.from_numpy(acts[s:e]), torch.arange(s, e))
            ref.update(acts[s:e], np.arange(s, e))
        assert np.array_equal(bits(am.activations), ref.vals), (tag, k, B)
        assert np.array_equal(am.sample_ids.numpy(), ref.ids), (tag, k, B)
        if tag in ("tiefree", "allneg"):
            assert np.array_equal(bits(am.activations), g[f"vals_{idx}"][-1])
            assert np.array_equal(am.sample_ids.numpy(), g[f"ids_{idx}"][-1])
        else:
            assert feq(oracle.bf16_to_f32(bits(am.activations)), oracle.bf16_to_f32(g[f"vals_{idx}"][-1]))


@pytest.mark.parametrize("C,k,B,N_", [(512, 20, 64, 1000), (2048, 100, 256, 1024), (37, 7, 50, 333), (5, 300, 64, 400)])
def test_random_streams_vs_oracle_both_modes(C, k, B, N_):
    rng = np.random.RandomState(C + k)
    acts = np.maximum(rng.randn(N_, C), 0).astype(np.float32)  # ReLU-like: many exact zeros + bf16 ties
    for mode, omode in (("aten", oracle.MODE_ATEN), ("total", oracle.MODE_TOTAL)):
        am = ActMax(n_collect=k, n_latents=C, tie_mode=mode)
        ref = oracle.ActMaxOracle(k, C, omode)
        for s in range(0, N_, B):
            e = min(N_, s + B)
            am.update(torch.from_numpy(acts[s:e]).to(DEV), torch.arange(s, e))
            ref.update(acts[s:e], np.arange(s, e))
        assert np.array_equal(bits(am.activations), ref.vals), mode
        assert np.array_equal(am.sample_ids.numpy(), ref.ids), mode


def test_total_mode_is_batch_invariant_and_shard_mergeable():
    rng = np.random.RandomState(3)
    N_, C, k = 3000, 256, 20
    acts = (rng.randint(0, 64, size=(N_, C)) / 8.0).astype(np.float32)  # tie-heavy
    x = torch.from_numpy(acts).to(DEV)
    results = []
    for B in (64, 256, 1000):
        am = ActMax(n_collect=k, n_latents=C, tie_mode="total")
        for s in range(0, N_, B):
            am.update(x[s : s + B], torch.arange(s, min(N_, s + B)))
        results.append((bits(am.activations), am.sample_ids.numpy()))
    for v, i in results[1:]:
        assert np.array_equal(v, results[0][0]) and np.array_equal(i, results[0][1])
    # 4 shards, merged with K4 == single stream
    bounds = [0, 700, 1500, 2300, N_]
    shards = []
    for r in range(4):
        am = ActMax(n_collect=k, n_latents=C, tie_mode="total")
        for s in range(bounds[r], bounds[r + 1], 128):
            e = min(bounds[r + 1], s + 128)
            am.update(x[s:e], torch.arange(s, e))
        shards.append(am)
    ov = torch.stack([shards[r].device_state()[0] for r in (0, 1, 3)])
    oi = torch.stack([shards[r].device_state()[1] for r in (0, 1, 3)])
    shards[2].merge_states(ov, oi)
    assert np.array_equal(bits(shards[2].activations), results[0][0])
    assert np.array_equal(shards[2].sample_ids.numpy(), results[0][1])
    ref = oracle.ActMaxOracle(k, C, oracle.MODE_TOTAL)
    ref.update(acts, np.arange(N_))
    assert np.array_equal(results[0][0], ref.vals) and np.array_equal(results[0][1], ref.ids)


# ------------------------------------------------------------------------------------------- K1/K2
CONV_SHAPES = [
    (1, 1, 1, 1), (1, 2, 1, 1), (1, 3, 1, 1), (1, 1, 1, 3), (3, 1, 1, 1), (1, 1, 1, 2),  # fewer than four floats in all
    (2, 3, 1, 1), (3, 5, 1, 3), (2, 4, 2, 2), (3, 7, 3, 3), (4, 8, 7, 7), (5, 6, 7, 7), (2, 16, 8, 8), (3, 5, 9, 9),
    (2, 9, 13, 13), (2, 8, 14, 14), (3, 3, 15, 17), (2, 4, 28, 28), (1, 3, 56, 56), (2, 2, 57, 59), (1, 2, 100, 103),
    (64, 256, 7, 7), (16, 64, 14, 14),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_reduce_vs_oracle(shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    x[0, 0, 0, 0] = float("nan")
    if shape[1] > 1:
        x[-1, 1].fill_(-3.0)
        x[-1, 1, -1, -1] = float("inf")
    xd = x.to(DEV)
    B, C = shape[:2]
    for name, code in (("max", N.SL_CONV_MAX), ("mean", N.SL_CONV_MEAN)):
        cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
        out = torch.empty((B, C), dtype=torch.float32, device=DEV)
        N.reduce_conv(xd, code, cand, out)
        want = oracle.agg_conv(x.numpy(), name)
        if name == "max":
            assert feq(out.cpu().numpy(), want), shape  # exact
            assert np.array_equal(bits(cand), oracle.f32_to_bf16(want)), shape
        else:
            np.testing.assert_allclose(out.cpu().numpy(), want, rtol=2e-6, atol=1e-6, equal_nan=True)
            d = np.abs(bits(cand).astype(np.int32) - oracle.f32_to_bf16(want).astype(np.int32))
            assert d.max() <= 1, shape  # <= 1 bf16 ulp
    # python-facing aggregators return the activation dtype on the host, like the reference
    assert feq(agg.aggregate_conv_max(xd).numpy(), oracle.agg_conv(x.numpy(), "max"))
    assert agg.aggregate_conv_max(xd).device.type == "cpu"


def test_conv_reduce_layouts_and_dtypes():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 32, 7, 7, generator=g)
    want = oracle.agg_conv(x.numpy(), "max")
    xd = x.to(DEV)
    for variant in (xd.contiguous(memory_format=torch.channels_last), xd[:, ::2], xd.transpose(2, 3), xd[:, :, 1:6, 2:5]):
        ref = oracle.agg_conv(variant.cpu().contiguous().numpy(), "max")
        out = torch.empty(variant.shape[:2], dtype=torch.float32, device=DEV)
        N.reduce_conv(variant, N.SL_CONV_MAX, None, out)
        assert feq(out.cpu().numpy(), ref)
    assert feq(agg.aggregate_conv_max(xd.contiguous(memory_format=torch.channels_last)).numpy(), want)
    # views with a size-1 spatial dimension: its stride is whatever the view had and must not be used
    y = torch.randn(4, 3, 1, 8, generator=g).to(DEV)
    for variant in (y.transpose(2, 3), y, y.transpose(2, 3)[:, :, ::2], y[:, :, :, ::3]):
        out = torch.empty(variant.shape[:2], dtype=torch.float32, device=DEV)
        N.reduce_conv(variant, N.SL_CONV_MAX, None, out)
        assert feq(out.cpu().numpy(), oracle.agg_conv(variant.cpu().contiguous().numpy(), "max"))
    for dt in (torch.float16, torch.bfloat16):
        xh = xd.to(dt)
        got = agg.aggregate_conv_max(xh)
        assert got.dtype == dt
        assert feq(got.float().numpy(), oracle.agg_conv(xh.float().cpu().numpy(), "max"))
        cand = torch.empty((6, 32), dtype=torch.bfloat16, device=DEV)
        N.reduce_conv(xh, N.SL_CONV_MAX, cand, None)
        assert np.array_equal(bits(cand), oracle.f32_to_bf16(oracle.agg_conv(xh.float().cpu().numpy(), "max")))


@pytest.mark.parametrize("shape", [(3, 4, 7, 7), (7, 12, 14, 14), (5, 8, 28, 28), (9, 8, 7, 7), (33, 64, 7, 7), (6, 10, 4, 4),
                                   (3, 16, 2, 2), (2, 6, 12, 12), (128, 512, 7, 7), (8, 100, 14, 14)])
def test_conv_reduce_special_values_and_partial_batches(shape):
    """NaN / +-inf anywhere (incl. +inf and -inf in one row: the sum-based NaN detector's false positive),
    rows of all -inf, and row counts that leave the last wave batch partly empty."""
    rng = np.random.RandomState(sum(shape))
    x = rng.randn(*shape).astype(np.float32)
    flat = x.reshape(shape[0] * shape[1], -1)
    R, S = flat.shape
    for r in rng.choice(R, size=min(R, 12), replace=False):
        kind = rng.randint(6)
        c = rng.randint(S)
        if kind == 0:
            flat[r, c] = np.nan
        elif kind == 1:
            flat[r, c] = np.inf
            flat[r, (c + 1) % S] = -np.inf  # sum is NaN, max is +inf
        elif kind == 2:
            flat[r, :] = -np.inf
        elif kind == 3:
            flat[r, S - 1] = np.nan  # last element of the row
        elif kind == 4:
            flat[r, 0] = -np.nan  # negative-signed NaN in the first element
        else:
            flat[r, c] = -np.inf
    xd = torch.from_numpy(x).to(DEV)
    B, C = shape[:2]
    for name, code in (("max", N.SL_CONV_MAX), ("mean", N.SL_CONV_MEAN)):
        cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
        out = torch.empty((B, C), dtype=torch.float32, device=DEV)
        N.reduce_conv(xd, code, cand, out)
        want = oracle.agg_conv(x, name)
        got = out.cpu().numpy()
        if name == "max":
            assert feq(got, want), shape
            assert np.array_equal(bits(cand), oracle.f32_to_bf16(want)), shape
        else:
            assert np.array_equal(np.isnan(got), np.isnan(want)), shape
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=3e-6, atol=1e-6)
            assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)])


HALF_SHAPES = [(2, 3, 1, 1), (3, 5, 1, 3), (4, 8, 7, 7), (5, 6, 7, 7), (3, 7, 3, 3), (2, 16, 8, 8), (3, 5, 9, 9), (2, 8, 14, 14),
               (3, 3, 15, 17), (2, 4, 28, 28), (1, 3, 56, 56), (2, 2, 57, 59), (1, 2, 100, 103), (33, 64, 7, 7), (16, 64, 14, 14),
               (8, 32, 28, 28), (9, 100, 14, 14), (2, 3, 40, 40)]


@pytest.mark.parametrize("shape", HALF_SHAPES)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_conv_reduce_half_precision_fast_paths(shape, dt):
    """fp16 / bf16 activations (half-precision models) on their own kernels: NCHW rows (rowreduce_h: 16-byte pieces of 8
    elements, rows starting on any 2-byte boundary), channels_last (8-byte loads of four components) and, through a
    misaligned view, the generic kernel — all three must agree with the oracle fed the same (rounded) values.  max is
    exact; mean accumulates in fp32 and is rounded once to the activation dtype, like torch's."""
    rng = np.random.RandomState(sum(shape) + (1 if dt == torch.float16 else 2))
    x = torch.from_numpy(rng.randn(*shape).astype(np.float32)).to(dt)
    flat = x.view(shape[0] * shape[1], -1)
    R, S = flat.shape
    for r in rng.choice(R, size=min(R, 10), replace=False):
        kind, c = rng.randint(6), rng.randint(S)
        if kind == 0:
            flat[r, c] = float("nan")
        elif kind == 1:
            flat[r, c] = float("inf")
            flat[r, (c + 1) % S] = -float("inf")
        elif kind == 2:
            flat[r, :] = -float("inf")
        elif kind == 3:
            flat[r, S - 1] = float("nan")
        elif kind == 4:
            flat[r, 0] = float("inf")
        else:
            flat[r, c] = -float("inf")
    xf = x.float().numpy()
    B, C = shape[:2]
    pad = torch.empty(x.numel() + 1, dtype=dt, device=DEV)
    pad[1:] = x.reshape(-1).to(DEV)
    variants = {"nchw": x.to(DEV), "channels_last": x.to(DEV).contiguous(memory_format=torch.channels_last),
                "misaligned": pad[1:].view(shape)}
    for name, code in (("max", N.SL_CONV_MAX), ("mean", N.SL_CONV_MEAN)):
        want = oracle.agg_conv(xf, name)
        if name == "mean":  # one rounding to the activation dtype (torch.mean of a half tensor)
            want = torch.from_numpy(want).to(dt).float().numpy()
        for tag, xd in variants.items():
            cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
            out = torch.empty((B, C), dtype=torch.float32, device=DEV)
            N.reduce_conv(xd, code, cand, out)
            got = out.cpu().numpy()
            if name == "max":
                assert feq(got, want), (tag, shape)
                assert np.array_equal(bits(cand), oracle.f32_to_bf16(want)), (tag, shape)
            else:
                assert np.array_equal(np.isnan(got), np.isnan(want)), (tag, shape)
                fin = np.isfinite(want)
                ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7  # one ulp of the activation dtype
                np.testing.assert_allclose(got[fin], want[fin], rtol=ulp, atol=1e-6)
                assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), (tag, shape)


DMA_HALF_SHAPES = [(128, 1024, 7, 7), (64, 512, 14, 14), (16, 512, 28, 28), (256, 2048, 3, 3), (128, 512, 9, 10), (512, 1024, 4, 4),
                   (256, 512, 8, 8), (16, 256, 32, 32), (131, 1001, 7, 7), (512, 1024, 2, 4),
                   # windows longer than a task's lanes (MULTI): 8 / 4 / 2 rows per task for odd S, even S, S % 4 == 0
                   (128, 512, 13, 13), (64, 512, 15, 14), (32, 512, 18, 18), (16, 512, 30, 30), (128, 1024, 9, 9),
                   # tasks of 4-16 KiB: one sixteen-instruction batch each
                   (32, 512, 17, 17), (16, 512, 27, 27), (16, 512, 31, 33), (8, 512, 42, 42)]


@pytest.mark.parametrize("shape", DMA_HALF_SHAPES)
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_conv_reduce_half_precision_dma_ring_special_values(shape, dt):
    """fp16 / bf16 NCHW inputs of >= 8 MB take `rowreduce_dma_kernel<T>` (round 3): aligned rows in one and in several steps
    (4 x 4, 8 x 8, 28 x 28, 32 x 32), unaligned rows with per-lane element masks (3 x 3, 7 x 7, 9 x 10, 14 x 14), and a row count
    the task rule excludes (131 x 1001 rows, not a multiple of eight: back to rowreduce_h).  NaN, +-inf, all -inf rows, values at a row's first and last
    element and in the tensor's last row; max exact (NaN propagates like torch.amax), mean within one ulp of the dtype."""
    rng = np.random.RandomState(sum(shape) + (3 if dt == torch.float16 else 4))
    B, C, H, W = shape
    assert B * C * H * W * 2 >= 8 << 20
    x = torch.from_numpy(rng.randn(B * C, H * W).astype(np.float32)).to(dt)
    R, S = x.shape
    rows = np.unique(np.concatenate([rng.choice(R, size=200, replace=False), [0, 1, R - 2, R - 1]]))
    for i, r in enumerate(rows):
        kind, c = i % 7, rng.randint(S)
        if kind == 0:
            x[r, c] = float("nan")
        elif kind == 1:
            x[r, c] = float("inf")
            x[r, (c + 1) % S] = -float("inf")
        elif kind == 2:
            x[r, :] = -float("inf")
        elif kind == 3:
            x[r, S - 1] = float("nan")
        elif kind == 4:
            x[r, 0] = float("inf")
        elif kind == 5:
            x[r, :] = -3.0
            x[r, S - 1] = 5.0  # the maximum in the row's last element: an edge piece of the next lane group
        else:
            x[r, :] = -3.0
            x[r, 0] = 7.0
    xd = x.view(shape).to(DEV)
    xf = x.float().numpy().reshape(shape)
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    for name, code in (("max", N.SL_CONV_MAX), ("mean", N.SL_CONV_MEAN)):
        want = oracle.agg_conv(xf, name)
        cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
        out = torch.empty((B, C), dtype=torch.float32, device=DEV)
        N.reduce_conv(xd, code, cand, out)
        got = out.cpu().numpy()
        if name == "max":
            assert feq(got, want), shape
            assert np.array_equal(bits(cand), oracle.f32_to_bf16(want)), shape
        else:
            want = torch.from_numpy(want).to(dt).float().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(want)), shape
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=ulp, atol=1e-6)
            assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), shape
    # absmax / absmean through the token entry point on the transposed view (B, T = S, F = C) with the token axis contiguous
    xt = xd.view(B, C, H * W).transpose(1, 2)
    for name in ("absmax", "absmean"):
        got = getattr(agg, f"aggregate_transformer_{name}")(xt).float().numpy()
        want = oracle.agg_tokens(np.ascontiguousarray(xf.reshape(B, C, H * W).transpose(0, 2, 1)), name)
        if name == "absmax":
            assert feq(got, want), (shape, name)
        else:
            want = torch.from_numpy(want).to(dt).float().numpy()
            assert np.array_equal(np.isnan(got), np.isnan(want)), (shape, name)
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=ulp, atol=1e-6)


@pytest.mark.parametrize("shape", [(64, 512, 13, 13), (64, 256, 15, 15), (64, 512, 9, 14), (32, 256, 21, 22), (63, 511, 13, 13),
                                   (32, 1024, 11, 11), (32, 256, 17, 17), (16, 256, 27, 27), (16, 256, 31, 33), (8, 256, 38, 39)])
def test_conv_reduce_fp32_dma_ring_unaligned_long_windows(shape):
    """fp32 rows that are not whole 16-byte pieces and longer than the 16 pieces of the single-step LDS-DMA path (odd maps of
    11 x 11 .. 15 x 15: four rows per task; even S up to 512: two rows per task) walk their window in steps with per-step
    element masks (`rowreduce_dma_kernel<..., MULTI>`); longer rows (17 x 17 .. 38 x 39) and 63 x 511 rows, which do not group
    into tasks, take the VGPR-load kernel.
    Special values at a row's first and last element, in the tensor's first and last rows; max exact, mean within 2e-6."""
    rng = np.random.RandomState(sum(shape))
    B, C, H, W = shape
    assert B * C * H * W * 4 >= 8 << 20
    x = torch.from_numpy(rng.randn(B * C, H * W).astype(np.float32))
    R, S = x.shape
    rows = np.unique(np.concatenate([rng.choice(R, size=200, replace=False), [0, 1, 2, 3, R - 4, R - 3, R - 2, R - 1]]))
    for i, r in enumerate(rows):
        kind, c = i % 7, rng.randint(S)
        if kind == 0:
            x[r, c] = float("nan")
        elif kind == 1:
            x[r, c] = float("inf")
            x[r, (c + 1) % S] = -float("inf")
        elif kind == 2:
            x[r, :] = -float("inf")
        elif kind == 3:
            x[r, S - 1] = float("nan")
        elif kind == 4:
            x[r, 0] = float("inf")
        elif kind == 5:
            x[r, :] = -3.0
            x[r, S - 1] = 5.0  # the maximum in the row's last element: shares a piece with the next row's first ones
        else:
            x[r, :] = -3.0
            x[r, 0] = 7.0
    xd = x.view(shape).to(DEV)
    xf = x.numpy().reshape(shape)
    for name, code in (("max", N.SL_CONV_MAX), ("mean", N.SL_CONV_MEAN), ("sum", N.SL_CONV_SUM)):
        want = oracle.agg_conv(xf, "mean" if name == "sum" else name)
        cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
        out = torch.empty((B, C), dtype=torch.float32, device=DEV)
        N.reduce_conv(xd, code, cand, out)
        got = out.cpu().numpy()
        if name == "max":
            assert feq(got, want), shape
            assert np.array_equal(bits(cand), oracle.f32_to_bf16(want)), shape
        else:
            if name == "sum":
                want = want * np.float32(S)
            assert np.array_equal(np.isnan(got), np.isnan(want)), (shape, name)
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=2e-5 if name == "sum" else 2e-6, atol=1e-5 * (S if name == "sum" else 1))
            assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), (shape, name)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_half_precision_full_size_equals_torch(dt):
    """ResNet-50 layer shapes at the bench batch size: the kernels' max equals torch.amax bit for bit in both layouts."""
    g = torch.Generator(device=DEV).manual_seed(3)
    for shape in ((256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7)):
        x = torch.randn(shape, device=DEV, generator=g).to(dt)
        want = x.amax((2, 3)).to(torch.bfloat16)
        for xd in (x, x.contiguous(memory_format=torch.channels_last)):
            cand = torch.empty(shape[:2], dtype=torch.bfloat16, device=DEV)
            N.reduce_conv(xd, N.SL_CONV_MAX, cand, None)
            assert torch.equal(cand, want), shape
        got = agg.aggregate_conv_mean(x)
        assert got.dtype == dt
        assert torch.allclose(got.float(), x.float().mean((2, 3)).cpu(), rtol=2.0 ** -7, atol=1e-6)


TOKEN_SHAPES = [(2, 10, 16), (3, 197, 24), (2, 5, 7), (4, 50, 768), (2, 197, 260), (1, 1, 4), (3, 33, 1000)]


@pytest.mark.parametrize("shape", TOKEN_SHAPES)
def test_token_reduce_vs_oracle(shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    x[0, 0, 0] = float("nan")
    xd = x.to(DEV)
    B, T, F = shape
    fns = {
        "mean": agg.aggregate_transformer_mean, "absmean": agg.aggregate_transformer_absmean,
        "max": agg.aggregate_transformer_max, "absmax": agg.aggregate_transformer_absmax,
    }
    for name, fn in fns.items():
        want = oracle.agg_tokens(x.numpy(), name)
        got = fn(xd).numpy()
        if "mean" in name:
            np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-6, equal_nan=True)
        else:
            assert feq(got, want), (shape, name)
    for pos in (0, -1, T // 2):
        got = agg.get_aggregate_transformer_special_token(pos)(xd).numpy()
        assert feq(got, oracle.agg_tokens(x.numpy(), "token", pos)), (shape, pos)
    # non-contiguous token tensor (e.g. a slice of a larger hidden state)
    if F >= 8:
        sl = xd[:, :, : F // 2]
        assert feq(agg.aggregate_transformer_max(sl).numpy(), oracle.agg_tokens(x[:, :, : F // 2].contiguous().numpy(), "max"))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T", [8, 24, 48, 144, 197, 576, 1024, 7, 13])
def test_token_reduce_transposed_half_precision(T, dt):
    """(B, T, F) views with the TOKEN axis contiguous (timm / ViT `patch_embed`: `flatten(2).transpose(1, 2)`) in fp16 /
    bf16 run rowreduce_h over rows of T elements.  T % 8 == 0 takes the whole-piece path, where lanes without a piece
    contribute a fill value: it must be neutral for every aggregator (round 2 filled absmax with |-inf| = +inf when
    T / 8 was not a power of two).  max / absmax exact, means within one ulp of the activation dtype."""
    B, F = 3, 40
    rng = np.random.RandomState(T + (0 if dt == torch.float16 else 1))
    base = torch.from_numpy(rng.randn(B, F, T).astype(np.float32)).to(dt)  # (B, F, T) contiguous
    base[0, 1, T // 2] = float("nan")
    base[1, 2, :] = -float("inf")
    base[2, 3, T - 1] = -float("inf")
    base[2, 5, 0] = float("inf")
    x = base.transpose(1, 2)  # (B, T, F) with st == 1, sf == T
    xd = base.to(DEV).transpose(1, 2)
    assert xd.stride() == (F * T, 1, T)
    ulp = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    xf = x.float().contiguous().numpy()
    for name in ("max", "absmax", "mean", "absmean"):
        got = getattr(agg, f"aggregate_transformer_{name}")(xd)
        assert got.dtype == dt
        got = got.float().numpy()
        want = oracle.agg_tokens(xf, name)
        if "mean" in name:
            want = torch.from_numpy(want).to(dt).float().numpy()  # one rounding to the activation dtype
            assert np.array_equal(np.isnan(got), np.isnan(want)), (name, T)
            fin = np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=ulp, atol=1e-6, err_msg=f"{name} T={T}")
            assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), (name, T)
        else:
            assert feq(got, want), (name, T, dt)


def test_aggregator_goldens(golden):
    g = golden("aggregators")
    for tag in ("x4a", "x4b", "x4c", "x4n"):
        x = torch.from_numpy(g[tag]).to(DEV)
        assert feq(agg.aggregate_conv_max(x).numpy(), g[f"{tag}_conv_max"])
        np.testing.assert_allclose(agg.aggregate_conv_mean(x).numpy(), g[f"{tag}_conv_mean"], rtol=3e-6, atol=2e-7, equal_nan=True)
    for tag in ("x3a", "x3b"):
        x = torch.from_numpy(g[tag]).to(DEV)
        assert feq(agg.aggregate_transformer_max(x).numpy(), g[f"{tag}_max"])
        assert feq(agg.aggregate_transformer_absmax(x).numpy(), g[f"{tag}_absmax"])
        assert feq(agg.get_aggregate_transformer_special_token(0)(x).numpy(), g[f"{tag}_tok0"])
        assert feq(agg.get_aggregate_transformer_special_token(-1)(x).numpy(), g[f"{tag}_tokm1"])
        np.testing.assert_allclose(agg.aggregate_transformer_mean(x).numpy(), g[f"{tag}_mean"], rtol=3e-6, atol=3e-7)
        np.testing.assert_allclose(agg.aggregate_transformer_absmean(x).numpy(), g[f"{tag}_absmean"], rtol=3e-6, atol=3e-7)
    with pytest.raises(ValueError, match="Input tensor should be 4D"):
        agg.aggregate_conv_max(torch.randn(2, 4, 8))
    with pytest.raises(ValueError, match="Input tensor should be 3D"):
        agg.aggregate_transformer_mean(torch.randn(2, 10, 16, 1))


# ------------------------------------------------------------------------------- fused hook path
@pytest.mark.parametrize("mode", ["aten", "total"])
def test_fused_hook_path_matches_oracle_resnet_like_shapes(mode):
    """ActMaxCache hooks on a conv stack: K1+K3 fused path vs oracle fed with the same activations."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(
        torch.nn.Conv2d(3, 32, 3, stride=2, padding=1), torch.nn.ReLU(),
        torch.nn.Conv2d(32, 64, 3, stride=2, padding=1), torch.nn.ReLU(),
        torch.nn.Conv2d(64, 128, 3, stride=2, padding=1), torch.nn.ReLU(),
    ).to(DEV).eval()
    layers = ["1", "3", "5"]
    k = 20
    cache = ActMaxCache(layers, agg.aggregate_conv_max, n_collect=k, tie_mode=mode)
    grabbed = {n: [] for n in layers}
    hooks = [model[int(n)].register_forward_hook(lambda m, i, o, n=n: grabbed[n].append(o.detach().cpu().numpy())) for n in layers]
    g = torch.Generator().manual_seed(1)
    sizes = [48, 48, 48, 48, 17]  # ragged last batch
    with torch.no_grad(), cache.hook_context(model):
        for b in sizes:
            model(torch.randn(b, 3, 56, 56, generator=g).to(DEV))
    for h in hooks:
        h.remove()
    for n in layers:
        C = grabbed[n][0].shape[1]
        ref = oracle.ActMaxOracle(k, C, oracle.MODE_ATEN if mode == "aten" else oracle.MODE_TOTAL)
        start = 0
        for a in grabbed[n]:
            ref.update(oracle.agg_conv(a, "max"), np.arange(start, start + a.shape[0]))
            start += a.shape[0]
        am = cache.cache[n]
        assert np.array_equal(bits(am.activations), ref.vals), (mode, n)
        assert np.array_equal(am.sample_ids.numpy(), ref.ids), (mode, n)
        assert cache.sample_idx_counter[n] == sum(sizes)


def test_token_hook_path_matches_oracle():
    class Blocks(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(24, 48)
            self.b = torch.nn.Linear(48, 40)

        def forward(self, x):
            return self.b(torch.relu(self.a(x)))

    torch.manual_seed(0)
    model = Blocks().to(DEV).eval()
    for fn, name, pos in ((agg.aggregate_transformer_max, "max", 0), (agg.get_aggregate_transformer_special_token(0), "token", 0)):
        cache = ActMaxCache(["a", "b"], fn, n_collect=7, tie_mode="aten")
        grabbed = {"a": [], "b": []}
        hooks = [getattr(model, n).register_forward_hook(lambda m, i, o, n=n: grabbed[n].append(o.detach().cpu().numpy())) for n in ("a", "b")]
        g = torch.Generator().manual_seed(2)
        with torch.no_grad(), cache.hook_context(model):
            for _ in range(4):
                model(torch.randn(9, 31, 24, generator=g).to(DEV))
        for h in hooks:
            h.remove()
        for n in ("a", "b"):
            ref = oracle.ActMaxOracle(7, grabbed[n][0].shape[2], oracle.MODE_ATEN)
            for i, a in enumerate(grabbed[n]):
                ref.update(oracle.agg_tokens(a, name, pos), np.arange(i * 9, i * 9 + 9))
            assert np.array_equal(bits(cache.cache[n].activations), ref.vals)
            assert np.array_equal(cache.cache[n].sample_ids.numpy(), ref.ids)


def test_custom_python_aggregator_goes_through_update():
    def my_l2_aggregate(t):  # a user-defined aggregator returning a host tensor, as the reference allows
        return t.flatten(2).norm(dim=-1).detach().cpu()

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3)).to(DEV).eval()
    cache = ActMaxCache(["0"], my_l2_aggregate, n_collect=4, tie_mode="aten")
    ref = oracle.ActMaxOracle(4, 8, oracle.MODE_ATEN)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad(), cache.hook_context(model):
        for i in range(3):
            x = torch.randn(5, 3, 8, 8, generator=g).to(DEV)
            ref.update(my_l2_aggregate(model(x)).numpy(), np.arange(5 * i, 5 * i + 5))
    am = cache.cache["0"]
    assert np.array_equal(bits(am.activations), ref.vals)
    assert np.array_equal(am.sample_ids.numpy(), ref.ids)


# ---------------------------------------------------------------------------------------------- K5
def test_gather_rows(golden):
    g = golden("pipeline")
    emb = torch.from_numpy(g["embeds"]).to(DEV)
    for name in ("0", "2"):
        out = N.gather_rows(emb, torch.from_numpy(g[f"ids_{name}"]))
        assert np.array_equal(out.cpu().numpy(), g[f"db_{name}"])
    rng = np.random.RandomState(0)
    for D in (5, 512, 1152):
        e = rng.randn(1000, D).astype(np.float32)
        ids = rng.randint(-1000, 1000, size=(64, 20))
        out = N.gather_rows(torch.from_numpy(e).to(DEV), torch.from_numpy(ids))
        assert np.array_equal(out.cpu().numpy(), oracle.gather_rows(e, ids))
    with pytest.raises(IndexError):
        N.gather_rows(emb, torch.tensor([emb.shape[0]]))


@pytest.mark.parametrize("D", [4, 20, 252, 256, 260, 512, 768, 1024, 1152, 1536, 1792, 2048, 2052, 30])
def test_gather_rows_every_width_class_and_the_sharded_form(D):
    """K5's wave-per-row kernel (round 6) has one instance per 256 floats of row width (1-8 pieces of 16 bytes per lane), a row-group
    size per instance, and two fall-backs (fewer than 64 ids, rows that are not whole 16-byte pieces or longer than 2 048 floats: the
    element-indexed kernel): every class against `emb[ids]`, negative ids wrapping, ragged id counts around the row-group sizes, and
    the sharded form (rows another shard holds come back as zeros; the shards' sum is the full gather)."""
    rng = np.random.RandomState(D)
    n_rows = 777
    e = rng.randn(n_rows, D).astype(np.float32)
    emb = torch.from_numpy(e).to(DEV)
    for n_ids in (1, 63, 64, 65, 67, 1000, 4099):
        ids = rng.randint(-n_rows, n_rows, size=n_ids)
        want = oracle.gather_rows(e, ids)
        out = N.gather_rows(emb, torch.from_numpy(ids))
        assert np.array_equal(out.cpu().numpy(), want), (D, n_ids)
        if n_ids in (65, 4099):
            total = np.zeros_like(want)
            for lo, hi in ((0, 300), (300, 301), (301, n_rows)):
                part = N.gather_rows_shard(emb[lo:hi].contiguous(), torch.from_numpy(ids), lo, n_rows).cpu().numpy()
                src = np.where(ids < 0, ids + n_rows, ids)
                mine = (src >= lo) & (src < hi)
                assert np.array_equal(part[mine], want[mine]) and not part[~mine].any(), (D, n_ids, lo)
                total += part
            assert np.array_equal(total, want)
    with pytest.raises(IndexError):
        N.gather_rows(emb, torch.arange(n_rows - 70, n_rows + 1))  # the wave kernel's path: 71 ids, the last one out of range


# ------------------------------------------------------------------------------------------ K6..K10
def test_scores_goldens(golden):
    g = golden("scores")
    t = lambda a: torch.from_numpy(g[a]).to(DEV)  # noqa: E731
    tol = dict(rtol=0, atol=1e-5)
    np.testing.assert_allclose(scores.similarity_score(t("sim_x"), t("sim_y")).cpu().numpy(), g["sim_xyT"], **tol)
    np.testing.assert_allclose(scores.similarity_score(t("sim_x"), t("sim_y2")).cpu().numpy(), g["sim_xy2"], **tol)
    np.testing.assert_allclose(scores.similarity_score(t("sim_x"), t("sim_y3")).cpu().numpy(), g["sim_rowwise"], **tol)
    np.testing.assert_allclose(scores.similarity_score(t("sim_xz"), t("sim_y")).cpu().numpy(), g["sim_xzyT"], **tol)
    np.testing.assert_allclose(scores.similarity_score(t("sim_xl"), t("sim_yl")).cpu().numpy(), g["sim_xlylT"], **tol)
    with pytest.raises(ValueError, match="x and y must have the same shape"):
        scores.similarity_score(torch.zeros(3, 4, device=DEV), torch.zeros(5, 6, device=DEV))
    np.testing.assert_allclose(scores.clarity_score(t("V")).cpu().numpy(), g["V_clarity"], **tol)
    np.testing.assert_allclose(scores.clarity_score(t("Vz")).cpu().numpy(), g["Vz_clarity"], **tol)
    r3 = scores.redundancy_score(t("V")[:, :15])
    assert r3.shape == (10,)
    np.testing.assert_allclose(r3.cpu().numpy(), g["V_redundancy3d"], **tol)
    r2 = scores.redundancy_score(t("V").mean(1))
    assert r2.shape == ()
    np.testing.assert_allclose(r2.cpu().numpy(), g["cones_redundancy"], **tol)
    # host tensors in -> host tensors out
    assert scores.clarity_score(torch.from_numpy(g["V"])).device.type == "cpu"


@pytest.mark.parametrize("Q,C,D", [(1, 10, 128), (130, 257, 100), (64, 768, 1152), (1000, 512, 512), (33, 70, 7)])
def test_cosine_gemm_vs_oracle(Q, C, D):
    rng = np.random.RandomState(Q + C + D)
    x = rng.randn(Q, D).astype(np.float32)
    y = (rng.randn(C, D) * rng.rand(C, 1) * 10).astype(np.float32)
    got = scores.similarity_score(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV)).cpu().numpy()
    assert got.shape == (Q, C)
    np.testing.assert_allclose(got, oracle.similarity(x, y), rtol=0, atol=1e-5)


def test_cosine_gemm_detects_transposes():
    """A = I-like probe with an asymmetric B: out[q][c] must be cos(x_q, y_c), not its transpose."""
    D = 64
    x = np.eye(40, D, dtype=np.float32)
    y = np.zeros((50, D), np.float32)
    for c in range(50):
        y[c, c % D] = 1.0
        y[c, (3 * c + 1) % D] += 2.0
    got = scores.similarity_score(torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV)).cpu().numpy()
    np.testing.assert_allclose(got, oracle.similarity(x, y), rtol=0, atol=1e-6)


@pytest.mark.parametrize("n,D", [(20, 512), (20, 1152), (100, 512), (7, 36), (2, 2048), (5, 2052), (9, 30), (33, 260)])
def test_clarity_all_layers_in_one_launch_vs_oracle(n, D):
    """`Lens.eval_clarity` over a concept_db dict (lens.py:391-419) = ONE `sl_clarity_multi` launch: layers with their own
    component counts (one of them empty, one a zero-norm slab) against the oracle layer by layer; odd / wide D take the two-pass
    kernel behind the same entry point."""
    from semanticlens_amd import Lens
    from helpers import FakeVLM

    rng = np.random.RandomState(n * 7 + D)
    layers = {f"l{i}": (rng.randn(c, n, D) * rng.rand(c, 1, 1) * 3).astype(np.float32) for i, c in enumerate((37, 1, 0, 300, 64))}
    layers["l1"][:] = 0.0  # F.normalize's eps path: every row has norm 0
    db = {k: torch.from_numpy(v).to(DEV) for k, v in layers.items()}
    got = Lens(FakeVLM(), device=DEV).eval_clarity(db)
    assert list(got) == list(db)
    with np.errstate(all="ignore"):
        for k, v in layers.items():
            want = oracle.clarity(v) if v.shape[0] else np.zeros((0,), np.float32)
            assert got[k].shape == (v.shape[0],) and got[k].is_cuda
            np.testing.assert_allclose(got[k].cpu().numpy(), want, rtol=0, atol=1e-5, equal_nan=True)
    # layer by layer through the public function: the same values to the last bit or two (another summation order inside a wave)
    for k, v in db.items():
        if v.shape[0] and n > 1:
            np.testing.assert_allclose(scores.clarity_score(v).cpu().numpy(), got[k].cpu().numpy(), rtol=0, atol=2e-6)
    # mixed (n, D) across layers cannot share a launch: the per-layer loop still answers
    mixed = {"a": db["l0"], "b": torch.randn(5, n + 1, D, device=DEV)}
    out = Lens(FakeVLM(), device=DEV).eval_clarity(mixed)
    np.testing.assert_allclose(out["b"].cpu().numpy(), oracle.clarity(mixed["b"].cpu().numpy()), rtol=0, atol=1e-5)


def test_clarity_redundancy_vs_oracle_larger():
    rng = np.random.RandomState(9)
    V = rng.randn(300, 20, 512).astype(np.float32)
    np.testing.assert_allclose(scores.clarity_score(torch.from_numpy(V).to(DEV)).cpu().numpy(), oracle.clarity(V), rtol=0, atol=1e-5)
    cones = V.mean(1)
    np.testing.assert_allclose(
        scores.redundancy_score(torch.from_numpy(cones).to(DEV)).cpu().numpy(), oracle.redundancy(cones), rtol=0, atol=1e-5
    )


def test_text_probes_golden(golden):
    from helpers import FakeVLM
    from semanticlens_amd.lens import _embed_text_probes

    g = golden("text_probes")
    fm = FakeVLM().to(DEV)
    queries = [str(q) for q in g["queries"]]
    templates = [str(t) for t in g["templates"]]
    for nq in (1, 3):
        for nt in (0, 1, 2):
            for bs in (0, 2):
                emb = _embed_text_probes(fm, queries[:nq], templates[:nt] or None, bs or None)
                assert np.array_equal(emb.cpu().numpy(), g[f"q{nq}_t{nt}_bs{bs}"]), (nq, nt, bs)


def test_redundancy_propagates_nan_like_torch_max():
    """scores.py:76-81: `.max(-1)` of the cosine matrix propagates NaN; a NaN embedding poisons the mean."""
    x = torch.randn(40, 32)
    assert torch.isfinite(scores.redundancy_score(x.to(DEV))).item()
    x[7, 3] = float("nan")
    assert torch.isnan(scores.redundancy_score(x.to(DEV))).item()


def test_image_probing_golden(golden):
    """lens.py:124-162 — a single image is used as is, several images are averaged; tensor and dict DBs."""
    from helpers import FakeVLM
    from semanticlens_amd.lens import Lens, image_probing

    g = golden("image_probes")
    fm = FakeVLM().to(DEV)
    imgs = torch.from_numpy(g["images"])
    db = torch.from_numpy(g["db"])
    db_dict = {"a": torch.from_numpy(g["db_a"]), "b": torch.from_numpy(g["db_b"])}
    lens = Lens(fm, device=DEV)
    for tag, query in (("one", imgs[0]), ("one_list", [imgs[1]]), ("three", [imgs[0], imgs[1], imgs[2]])):
        for where in ("cpu", DEV):  # results come back on the DB's device, like the reference's `_probe`
            got = image_probing(fm, query, db.to(where))
            assert got.device.type == torch.device(where).type and got.shape == g[f"{tag}_tensor"].shape
            np.testing.assert_allclose(got.cpu().numpy(), g[f"{tag}_tensor"], rtol=0, atol=1e-5)
            res = lens.image_probing(query, {k: v.to(where) for k, v in db_dict.items()})
            assert list(res) == ["a", "b"]
            for k in res:
                assert res[k].shape == g[f"{tag}_{k}"].shape
                np.testing.assert_allclose(res[k].cpu().numpy(), g[f"{tag}_{k}"], rtol=0, atol=1e-5)
    # the averaged query embedding itself is bit-equal to the reference's (integer projection, one host division)
    from semanticlens_amd.lens import _embed_image_probe

    assert np.array_equal(_embed_image_probe(fm, [imgs[0], imgs[1], imgs[2]]).cpu().numpy(), g["three_query_embed"])


# ---------------------------------------------------------------------------------------------- K9
def test_polysemanticity_goldens(golden):
    """scores.py:131-185 incl. the fallback rows (one cluster / a cluster of one sample)."""
    g = golden("scores")
    for key in ("P", "P10"):
        got = scores.polysemanticity_score(torch.from_numpy(g[key]).to(DEV))
        assert got.dtype == torch.float64 and got.shape == (g[key].shape[0],)
        np.testing.assert_allclose(got.cpu().numpy(), g[f"{key}_poly"], rtol=0, atol=1e-5)
    assert scores.polysemanticity_score(torch.from_numpy(g["P10"])).device.type == "cpu"


def test_polysemanticity_goldens_any_n_clusters(golden):
    """n_clusters = 3 / 4 / 6 against outputs of the reference itself (tests/golden/scores_k.npz): random rows, three- and
    four-blob rows, rows of duplicated points (empty clusters relocated, the fallback branch)."""
    g = golden("scores_k")
    for k in (3, 4, 6):
        got = scores.polysemanticity_score(torch.from_numpy(g["P"]).to(DEV), n_clusters=k)
        np.testing.assert_allclose(got.cpu().numpy(), g[f"poly_k{k}"], rtol=0, atol=1e-5)


def assert_polysemanticity_matches(got, V, tag="", n_clusters=2):
    """Every component must reproduce scikit-learn's clustering — no tolerated minority.

    The reference is deterministic (scores.py:167, random_state=123).  Rows scored from the cluster centres are
    fp64 on both sides: a different clustering moves the score by >> 1e-9, so agreement within 1e-9 shows the same
    clustering was chosen.  Rows that take the fallback branch (scores.py:178-184) are computed by the reference
    through the fp32 ``clarity_score``: 1e-6.  Observed on MI355X (tools/k9_rate.py): 100 % of 2 400 components.
    """
    want, fallback = oracle.polysemanticity(V, n_clusters=n_clusters, return_fallback=True)
    diff = np.abs(got - want)
    print(f"K9 {tag}: {len(want)} components, {int(fallback.sum())} fallback rows, max |diff| centres "
          f"{diff[~fallback].max() if (~fallback).any() else 0:.2e}, fallback {diff[fallback].max() if fallback.any() else 0:.2e}")
    bad = np.nonzero((diff > 1e-9) & ~fallback)[0]
    assert bad.size == 0, (tag, "clustering differs from scikit-learn's for components", bad[:10], diff[bad[:10]])
    bad = np.nonzero((diff > 1e-6) & fallback)[0]
    assert bad.size == 0, (tag, "fallback rows differ", bad[:10], diff[bad[:10]])


@pytest.mark.parametrize("C,n,D,kind", [(256, 20, 512, "random"), (256, 20, 512, "blobs"), (128, 10, 64, "random"),
                                        (64, 33, 100, "blobs"), (64, 100, 32, "random"), (96, 20, 1152, "blobs3"),
                                        (512, 20, 512, "weak"), (256, 5, 16, "random"), (128, 3, 8, "random")])
def test_polysemanticity_vs_sklearn_oracle(C, n, D, kind):
    """The Gram-space restatement of sklearn's KMeans picks the same clustering as sklearn itself
    (oracle.polysemanticity calls scikit-learn) for EVERY component."""
    rng = np.random.RandomState(C + n + D)
    V = rng.randn(C, n, D).astype(np.float32)
    if kind.startswith("blobs"):
        nb = 3 if kind == "blobs3" else 2
        centers = rng.randn(C, nb, D).astype(np.float32) * 2
        assign = rng.randint(0, nb, size=(C, n))
        V = centers[np.arange(C)[:, None], assign] + 0.5 * V
    elif kind == "weak":  # barely separated blobs: many competing local optima across the 10 inits
        centers = rng.randn(C, 2, D).astype(np.float32) * 0.15
        V = centers[np.arange(C)[:, None], rng.randint(0, 2, size=(C, n))] + V
    got = scores.polysemanticity_score(torch.from_numpy(V).to(DEV)).cpu().numpy()
    assert_polysemanticity_matches(got, V, kind)
    assert np.all(got >= -1e-6) and np.all(got <= 2 + 1e-6)


@pytest.mark.parametrize("C,n,D,k,kind", [(96, 20, 64, 3, "random"), (96, 20, 64, 4, "blobs"), (64, 12, 16, 5, "random"),
                                          (64, 30, 8, 3, "blobs"), (48, 8, 4, 6, "dup"), (24, 150, 32, 2, "blobs"),
                                          (16, 200, 24, 3, "random"), (64, 20, 512, 16, "random")])
def test_polysemanticity_any_n_clusters_vs_sklearn(C, n, D, k, kind):
    """scores.py:132,167: `n_clusters` is forwarded to scikit-learn.  The general kernel (any k up to 16, n up to 1024,
    relocation of several empty clusters, k-means++ with 2 + int(ln k) trials) must pick scikit-learn's clustering for
    every component; "dup" duplicates points so that empty clusters and the fallback rows occur."""
    rng = np.random.RandomState(C + n + D + k)
    V = rng.randn(C, n, D).astype(np.float32)
    if kind == "blobs":
        cen = rng.randn(C, k, D).astype(np.float32) * 2
        V = cen[np.arange(C)[:, None], rng.randint(0, k, size=(C, n))] + 0.4 * V
    if kind == "dup":
        V[:, n // 2:] = V[:, : n - n // 2]
        V[:, :3] = V[:, :1]
    got = scores.polysemanticity_score(torch.from_numpy(V).to(DEV), n_clusters=k).cpu().numpy()
    assert_polysemanticity_matches(got, V, f"{kind} k={k}", n_clusters=k)
    with pytest.raises(ValueError):  # sklearn: n_samples should be >= n_clusters
        scores.polysemanticity_score(torch.from_numpy(V[:, :2]).to(DEV), n_clusters=3)


def test_polysemanticity_chunks_components_and_reports_limits(monkeypatch):
    """The reference loops over components on the host and has no size limit (scores.py:167).  The device kernels take
    <= 65 535 components per launch inside a workspace budget: the wrapper chunks the component axis — same scores as one
    launch — and the two limits that remain (n_samples > 1024, n_clusters > 16) are ValueErrors, not native failures."""
    rng = np.random.RandomState(5)
    V = torch.from_numpy(rng.randn(150, 12, 16).astype(np.float32)).to(DEV)
    Vdup = V.clone()
    Vdup[:20, 1:] = Vdup[:20, :1]  # all-duplicate rows (dead components: every reference sample the same)
    for X in (V, Vdup):
        for k in (2, 3):
            monkeypatch.delenv("SL_POLY_WS_GB", raising=False)
            whole = scores.polysemanticity_score(X, n_clusters=k)
            whole_raw = scores.polysemanticity_score(X, n_clusters=k, replace_empty_clusters=False)
            monkeypatch.setenv("SL_POLY_WS_GB", "1e-4")  # ~100 KB: a handful of components per launch
            assert torch.equal(scores.polysemanticity_score(X, n_clusters=k), whole)
            monkeypatch.setattr(N, "POLYK_MAX_COMPONENTS", 7)
            assert torch.equal(scores.polysemanticity_score(X, n_clusters=k, replace_empty_clusters=False), whole_raw)
            monkeypatch.undo()
    # every centre of an all-duplicate component coincides: polysemanticity 0 without the fallback, for any k
    raw = scores.polysemanticity_score(Vdup[:20], n_clusters=4, replace_empty_clusters=False).cpu().numpy()
    np.testing.assert_allclose(raw, 0.0, atol=1e-9)
    with pytest.raises(ValueError, match="n_clusters=17"):
        scores.polysemanticity_score(torch.zeros(2, 40, 4, device=DEV), n_clusters=17)
    with pytest.raises(ValueError, match="n_samples=1025"):
        scores.polysemanticity_score(torch.zeros(1, 1025, 4, device=DEV), n_clusters=3)
    assert scores.polysemanticity_score(torch.zeros(0, 5, 4, device=DEV)).shape == (0,)


def test_probe_dict_uses_one_native_call_and_matches_per_layer():
    from semanticlens_amd.lens import _probe

    g = torch.Generator(device=DEV).manual_seed(11)
    q = torch.randn(37, 256, device=DEV, generator=g)
    db = {"a": torch.randn(50, 256, device=DEV, generator=g), "b": torch.randn(300, 256, device=DEV, generator=g),
          "c": torch.randn(1, 256, device=DEV, generator=g)}
    got = _probe(q, db)
    for k, v in db.items():
        assert got[k].shape == (37, v.shape[0]) and got[k].is_contiguous()
        np.testing.assert_allclose(got[k].cpu().numpy(), oracle.similarity(q.cpu().numpy(), v.cpu().numpy()), rtol=0, atol=1e-5)
    # a layer that hits a shape quirk (C == Q -> row-wise cosine) makes the whole dict go layer by layer
    db["quirk"] = torch.randn(37, 256, device=DEV, generator=g)
    got = _probe(q, db)
    assert got["quirk"].shape == (37,) and got["b"].shape == (37, 300)
    # host tensors in -> host tensors out
    host = _probe(q.cpu(), {k: v.cpu() for k, v in db.items() if k != "quirk"})
    assert all(t.device.type == "cpu" for t in host.values())


# ---------------------------------------------------------------------------------------------- K6 variants
def test_fused_multi_layer_probe_equals_per_layer_similarity():
    """sl_similarity_multi (all layers in ONE GEMM launch, routing epilogue) == sl_similarity layer by layer, bit for
    bit in both arithmetic modes; ragged layer sizes, an empty layer, column counts that are not tile multiples."""
    torch.manual_seed(5)
    q = torch.randn(333, 136, device=DEV)
    ys = [torch.randn(c, 136, device=DEV) for c in (768, 5, 0, 130, 257, 1024)]
    try:
        for mode in ("bf16x3", "f32"):
            N.set_gemm_mode(mode)
            outs = N.similarity_multi(q, ys)
            assert outs is not None and len(outs) == len(ys)
            for y, o in zip(ys, outs):
                assert o.shape == (333, y.shape[0])
                if y.shape[0]:
                    assert torch.equal(o, N.similarity(q, y)), (mode, y.shape)
                    ref = oracle.similarity(q.cpu().numpy(), y.cpu().numpy())
                    assert np.abs(o.cpu().numpy() - ref).max() < 1e-5
        # large enough that the fp32 mode gathers the layers into one operand for the 256 x 256 8-phase kernel
        # (16 x 13 tiles fused; no single layer would fill half the chip) — still bit-identical to layer by layer
        q2 = torch.randn(4096, 256, device=DEV)
        ys2 = [torch.randn(c, 256, device=DEV) for c in (768, 1024, 1300, 5)]
        for mode in ("bf16x3", "f32"):
            N.set_gemm_mode(mode)
            for y, o in zip(ys2, N.similarity_multi(q2, ys2)):
                assert torch.equal(o, N.similarity(q2, y)), (mode, y.shape)
                assert np.abs(o.cpu().numpy() - oracle.similarity(q2.cpu().numpy(), y.cpu().numpy())).max() < 1e-5
    finally:
        N.set_gemm_mode(None)
    # a layer that would hit a shape quirk makes the native call decline (the caller goes layer by layer)
    assert N.similarity_multi(q, [torch.randn(333, 136, device=DEV)]) is None


def test_gemm_tile_variants_are_bit_identical(tmp_path):
    """The 128x128 register-staged, the 256x128 LDS-DMA staged, the 256x256 ping-pong, the 256x256 8-phase, the 160x256
    four-wave and the 64x64 / 128x128 ring (round 3: "160", "64", "1280") split-bf16 kernels accumulate every output element in
    the same order: same bits (the variant is
    latched per process, hence subprocesses).  Forcing a variant sends EVERY shape through it, ragged and tiny ones included."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, torch
sys.path.insert(0, sys.argv[2])
from semanticlens_amd import _native as N
torch.manual_seed(0)
outs = []
for (q, c, d) in [(1000, 768, 1152), (130, 257, 64), (129, 128, 72), (1, 5, 4096), (300, 301, 104), (2048, 1024, 512), (513, 259, 200), (161, 257, 20), (320, 512, 32)]:  # the last two: a single k-tile
    x = torch.randn(q, d, device="cuda:0"); y = torch.randn(c, d, device="cuda:0")
    outs.append(N.similarity(x, y).cpu())
# full-size probe (race screen for the LDS-DMA pipeline): three runs, order-sensitive checksums of the raw bits
x = torch.randn(10000, 1152, device="cuda:0"); y = torch.randn(9216, 1152, device="cuda:0")
wts = torch.arange(1, 9217, device="cuda:0", dtype=torch.int64)
for _ in range(3):
    o = N.similarity(x, y).view(torch.int32).to(torch.int64)
    outs.append(torch.stack([o.sum(), (o * wts).sum(), (o.sum(1) * torch.arange(1, 10001, device="cuda:0")).sum()]).cpu())
torch.save(outs, sys.argv[1])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tile in ("128", "256", "8", "160", "64", "1280"):
        out = tmp_path / f"g3_{tile}.pt"
        subprocess.run([sys.executable, "-c", code, str(out), root], check=True, env=dict(os.environ, SL_OPTIONS=f"g3_tile={tile}"))
        res[tile] = torch.load(out)
    for other in ("256", "8", "160", "64", "1280"):
        for a, b in zip(res["128"], res[other]):
            assert torch.equal(a, b), (other, tuple(a.shape))
    # the fp32-MFMA mode has two tile variants too (128 x 128 register-staged, 256 x 256 8-phase): same bits
    f32 = {}
    for tile in ("128", "8"):
        out = tmp_path / f"f32_{tile}.pt"
        subprocess.run([sys.executable, "-c", code, str(out), root], check=True,
                       env=dict(os.environ, SL_GEMM_MODE="f32", SL_OPTIONS=f"f32_tile={tile}"))
        f32[tile] = torch.load(out)
    for a, b in zip(f32["128"], f32["8"]):
        assert torch.equal(a, b), ("f32", tuple(a.shape))


def test_linear_epilogues_are_bit_identical_across_tile_variants_on_ragged_shapes(tmp_path):
    """The encoder's GEMM epilogues (bias, GELU -> split output, residual added in place, scattered rows + positional table)
    through every split-bf16 tile variant, on shapes whose last row and column tiles are partial for each of them
    (12 750 x 700: 80 tiles of 160 rows with 110 valid in the last, 50 of 256 with 206; 700 columns = 2.73 tiles): same bits,
    and equal to the fp64 product within split-bf16 accuracy."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, torch
sys.path.insert(0, sys.argv[2])
from semanticlens_amd import _native as N
torch.manual_seed(0)
dev = "cuda:0"
outs = []
for (M, Nn, K) in [(12750, 700, 96), (321, 1030, 40), (12750, 768, 64)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) * 0.1; b = torch.randn(Nn, device=dev)
    sx, sw = N.Split.of(x), N.Split.of(w)
    res = torch.randn(M, Nn, device=dev)
    o1 = N.linear3(sx, sw, b)                                     # bias -> fp32
    o2 = res.clone(); N.linear3(sx, sw, b, residual=o2, out=o2)   # bias + residual in place
    sp = N.Split(M, Nn, dev); N.linear3(sx, sw, b, act=N.SL_ACT_GELU, out_split=sp)  # bias + GELU -> split
    o3 = sp.hi.float() + sp.lo.float()
    ref = (x.double() @ w.double().T + b.double())
    outs += [o1.cpu(), o2.cpu(), o3.cpu(), (o1.double() - ref).abs().max().cpu(), ref.abs().max().cpu()]
torch.save(outs, sys.argv[1])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tile in ("128", "256", "8", "160", "64", "1280"):
        out = tmp_path / f"lin_{tile}.pt"
        subprocess.run([sys.executable, "-c", code, str(out), root], check=True, env=dict(os.environ, SL_OPTIONS=f"g3_tile={tile}"))
        res[tile] = torch.load(out)
    for other in ("256", "8", "160", "64", "1280"):
        for i, (a, b) in enumerate(zip(res["128"], res[other])):
            assert torch.equal(a, b), (other, i, tuple(a.shape))
    for i in range(0, len(res["128"]), 5):
        err, scale = float(res["128"][i + 3]), float(res["128"][i + 4])
        assert err <= 2e-6 * max(scale, 1.0) * 4, (i, err, scale)


def test_column_strip_split_keeps_every_bit(tmp_path):
    """A GEMM whose last column tile is partial may be cut at the last full tile (gemm_bf16x3.hpp `strip_split_columns`: the big
    kernel on the first 256 n columns, the strip's own kernel on the rest with the epilogue shifted): same bits as the uncut GEMM
    (option `g3_strip_off`) for every epilogue — plain, cosine routing over several layers, bias, residual in place, GELU -> split output,
    scattered rows + positional table — at shapes the cost model does cut (so400m's N = 1152 and 4304 at 64 images, 3 x 257 layers)."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, torch
sys.path.insert(0, sys.argv[2])
from semanticlens_amd import _native as N
torch.manual_seed(0)
dev = "cuda:0"
outs = []
for (M, Nn, K) in [(16384, 1152, 96), (16384, 4304, 64), (16130, 1100, 40), (16384, 3456, 40)]:  # the last: cut one full tile early
    x = torch.randn(M, K, device=dev); w = torch.randn(Nn, K, device=dev) * 0.1; b = torch.randn(Nn, device=dev)
    sx, sw = N.Split.of(x), N.Split.of(w)
    res = torch.randn(M, Nn, device=dev)
    o1 = N.linear3(sx, sw, b)
    o2 = res.clone(); N.linear3(sx, sw, b, residual=o2, out=o2)
    sp = N.Split(M, Nn, dev); N.linear3(sx, sw, b, act=N.SL_ACT_GELU, out_split=sp)
    outs += [o1.cpu(), o2.cpu(), sp.hi.cpu().view(torch.int16), sp.lo.cpu().view(torch.int16)]
    if M % 64 == 0:  # rows scattered behind a class-token row per group of 64, positional rows added
        T = 65
        tab = torch.randn(T, Nn, device=dev)
        o4 = torch.zeros((M // 64) * T, Nn, device=dev)
        N.linear3(sx, sw, b, out=o4, scatter=(64, T, 1), rowadd=tab)
        outs.append(o4.cpu())
q = torch.randn(16384, 72, device=dev)
outs.append(N.similarity(q, torch.randn(1152, 72, device=dev)).cpu())
layers = [torch.randn(c, 72, device=dev) for c in (257, 640, 255)]  # 1152 columns routed to three outputs
outs += [o.cpu() for o in N.similarity_multi(q, layers)]
torch.save(outs, sys.argv[1])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for strip in ("1", "0"):
        out = tmp_path / f"strip_{strip}.pt"
        subprocess.run([sys.executable, "-c", code, str(out), root], check=True, env=dict(os.environ, SL_OPTIONS=f"g3_strip_off={1 - int(strip)}"))
        res[strip] = torch.load(out)
    assert len(res["1"]) == len(res["0"]) >= 15
    for i, (a, b) in enumerate(zip(res["1"], res["0"])):
        assert torch.equal(a, b), (i, tuple(a.shape))


@pytest.mark.parametrize("rpg,gs,ro", [(1, 3, 1), (2, 2, 0), (7, 9, 2), (49, 50, 1), (197, 200, 3), (1000, 1001, 1)])
def test_linear_row_scatter_maps_rows_exactly(rpg, gs, ro):
    """The scattering epilogue (patch embeddings land behind the class token of their image and take their row of the positional
    table) computes row / rpg and row % rpg with a multiply-high + correction instead of divisions: every output row is the one
    the formula names, every other row of the buffer stays untouched; both arithmetic modes."""
    torch.manual_seed(rpg)
    Ms = 3000
    xs, ws = torch.randn(Ms, 64, device=DEV), torch.randn(96, 64, device=DEV) * 0.1
    groups = -(-Ms // rpg)
    tab = torch.randn(ro + rpg, 96, device=DEV)
    r = torch.arange(Ms, device=DEV)
    rows = (r // rpg) * gs + ro + r % rpg
    want = xs.double() @ ws.double().T + tab[ro + r % rpg].double()
    untouched = torch.ones(groups * gs + ro + rpg, dtype=torch.bool, device=DEV)
    untouched[rows] = False
    for mode in ("bf16x3", "f32"):
        o = torch.full((groups * gs + ro + rpg, 96), 7.0, device=DEV)
        if mode == "bf16x3":
            N.linear3(N.Split.of(xs), N.Split.of(ws), None, out=o, scatter=(rpg, gs, ro), rowadd=tab)
        else:
            N.linear(xs, ws, None, out=o, scatter=(rpg, gs, ro), rowadd=tab)
        assert float((o[rows].double() - want).abs().max()) < 1e-4, (mode, rpg)
        assert bool((o[untouched] == 7.0).all()), (mode, rpg)


def test_aggregator_goldens_half_precision(golden):
    """The reference's own outputs on fp16 / bf16 activations (tests/golden/aggregators_half.npz, CPU torch): result dtype =
    activation dtype; max / absmax exact; means within one ulp of that dtype (fp32 summation order differs)."""
    g = golden("aggregators_half")
    for dname, dt, ulp in (("f16", torch.float16, 2.0 ** -10), ("bf16", torch.bfloat16, 2.0 ** -7)):
        for tag in ("h4a", "h4b", "h4c"):
            x = torch.from_numpy(g[f"{tag}_{dname}"]).to(dt).to(DEV)
            for layout in (x, x.contiguous(memory_format=torch.channels_last)):
                got = agg.aggregate_conv_max(layout)
                assert got.dtype == dt and got.device.type == "cpu"
                assert feq(got.float().numpy(), g[f"{tag}_{dname}_aggregate_conv_max"])
                got = agg.aggregate_conv_mean(layout)
                assert got.dtype == dt
                np.testing.assert_allclose(got.float().numpy(), g[f"{tag}_{dname}_aggregate_conv_mean"], rtol=ulp, atol=1e-6)
        for tag in ("h3a", "h3b"):
            x = torch.from_numpy(g[f"{tag}_{dname}"]).to(dt).to(DEV)
            for fn in ("aggregate_transformer_max", "aggregate_transformer_absmax"):
                got = getattr(agg, fn)(x)
                assert got.dtype == dt
                assert feq(got.float().numpy(), g[f"{tag}_{dname}_{fn}"]), (tag, dname, fn)
            for fn in ("aggregate_transformer_mean", "aggregate_transformer_absmean"):
                got = getattr(agg, fn)(x)
                assert got.dtype == dt
                np.testing.assert_allclose(got.float().numpy(), g[f"{tag}_{dname}_{fn}"], rtol=ulp, atol=1e-6, err_msg=f"{tag} {dname} {fn}")
