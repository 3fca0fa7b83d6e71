"""Parity at BASELINE.json's full sizes (configs[1]: ResNet-50 layer2/3/4 at batch 256, 50k samples, k=20/100)
through the oracle where it finishes in seconds and through size-independent properties otherwise."""
import numpy as np
import pytest
import torch

import oracle
from semanticlens_amd import _native as N
from semanticlens_amd import scores
from semanticlens_amd.component_visualization.activation_caching import ActMax

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def bits(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("shape", [(256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7)])
def test_reduce_full_batch_shapes(shape):
    g = torch.Generator(device=DEV).manual_seed(shape[1])
    x = torch.randn(*shape, device=DEV, generator=g).relu_()
    B, C = shape[:2]
    cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
    out = torch.empty((B, C), dtype=torch.float32, device=DEV)
    N.reduce_conv(x, N.SL_CONV_MAX, cand, out)
    # independent device reference (torch.amax) on the whole tensor + the CPU oracle on a slab
    assert torch.equal(out, x.flatten(2).amax(-1))
    assert torch.equal(cand, out.to(torch.bfloat16))
    sl = x[:16].cpu().numpy()
    assert np.array_equal(out[:16].cpu().numpy(), oracle.agg_conv(sl, "max"))
    N.reduce_conv(x, N.SL_CONV_MEAN, cand, out)
    torch.testing.assert_close(out, x.flatten(2).mean(-1), rtol=3e-6, atol=1e-6)
    # idempotence / linearity-style property: max over a permuted spatial axis is unchanged
    perm = torch.randperm(shape[2] * shape[3], device=DEV)
    xp = x.flatten(2)[:, :, perm].reshape(shape).contiguous()
    out2 = torch.empty_like(out)
    N.reduce_conv(xp, N.SL_CONV_MAX, None, out2)
    N.reduce_conv(x, N.SL_CONV_MAX, None, out)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("k", [20, 100])
def test_stream_50k_samples_layer4_width(k):
    """N = 50,176 samples x C = 2048 components: total mode vs the oracle, batch-size invariance, and the
    aten mode's VALUES equal the total mode's (they may differ only in which tied sample is listed)."""
    C, N_, B = 2048, 50176, 256
    g = torch.Generator(device=DEV).manual_seed(k)
    ref = oracle.ActMaxOracle(k, C, oracle.MODE_TOTAL)
    am = ActMax(k, C, tie_mode="total")
    am_big = ActMax(k, C, tie_mode="total")
    am_aten = ActMax(k, C, tie_mode="aten")
    pend = []
    for s in range(0, N_, B):
        a = torch.randn(B, C, device=DEV, generator=g).relu_()
        ids = torch.arange(s, s + B)
        am.update(a, ids)
        am_aten.update(a, ids)
        pend.append(a)
        if len(pend) == 4:  # the same stream in batches of 1024
            am_big.update(torch.cat(pend), torch.arange(s + B - 4 * B, s + B))
            pend = []
        ref.update(a.cpu().numpy(), ids.numpy())
    assert np.array_equal(bits(am.activations), ref.vals)
    assert np.array_equal(am.sample_ids.numpy(), ref.ids)
    assert np.array_equal(bits(am_big.activations), ref.vals) and np.array_equal(am_big.sample_ids.numpy(), ref.ids)
    assert np.array_equal(am_aten.activations.float().numpy(), am.activations.float().numpy())
    ids = am_aten.sample_ids.numpy()
    assert ids.min() >= 0 and ids.max() < N_
    assert all(len(set(r)) == k for r in ids[:64])


def test_probing_full_size_against_fp64_on_device():
    """configs[3] shapes: Q = 10,000 x C = 768 x D = 1152 (one of the 12 layers), checked against a float64
    cosine computed independently on the device."""
    g = torch.Generator(device=DEV).manual_seed(7)
    q = torch.randn(10000, 1152, device=DEV, generator=g)
    db = torch.randn(768, 1152, device=DEV, generator=g) * torch.rand(768, 1, device=DEV, generator=g) * 10
    got = scores.similarity_score(q, db)
    want = torch.nn.functional.normalize(q.double(), dim=-1) @ torch.nn.functional.normalize(db.double(), dim=-1).T
    assert got.shape == (10000, 768)
    assert (got.double() - want).abs().max().item() < 1e-5
    sub = oracle.similarity(q[:64].cpu().numpy(), db.cpu().numpy())
    np.testing.assert_allclose(got[:64].cpu().numpy(), sub, rtol=0, atol=1e-5)


def test_scores_full_layer_sizes():
    """concept_db of ResNet-50 layer4 at k = 20, D = 512: (2048, 20, 512)."""
    g = torch.Generator(device=DEV).manual_seed(9)
    V = torch.randn(2048, 20, 512, device=DEV, generator=g)
    clar = scores.clarity_score(V)
    Vn = torch.nn.functional.normalize(V.double(), dim=-1)
    want = ((Vn.mean(1) ** 2).sum(-1) - 1 / 20) / 19 * 20
    assert (clar.double() - want).abs().max().item() < 1e-5
    red = scores.redundancy_score(V.mean(1))
    cn = torch.nn.functional.normalize(V.mean(1).double(), dim=-1)
    sims = cn @ cn.T - 2 * torch.eye(2048, device=DEV, dtype=torch.float64)
    assert abs(red.item() - sims.max(-1).values.mean().item()) < 1e-5
    poly = scores.polysemanticity_score(V)
    assert poly.shape == (2048,) and poly.dtype == torch.float64
    assert torch.all(poly >= 0) and torch.all(poly <= 2)
    np.testing.assert_allclose(poly[:24].cpu().numpy(), oracle.polysemanticity(V[:24].cpu().numpy()), rtol=0, atol=1e-5)
