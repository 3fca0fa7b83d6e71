"""colreduce2 (K2 / channels_last K1, round 4) against the oracle over random shapes that reach every (LPR, waves) variant:
F a multiple of the 16-byte piece up to 4 352, 1-700 rows, fp32 / fp16 / bf16, every token aggregator and the conv
aggregators on channels_last maps, NaN / +-inf / -0.0 planted, row slices (t_begin / t_end of the special-token path), and the
forced wave counts.  Synchronises after every call and prints the case first.  python tools/fuzz_k2.py [seed] [cases]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oracle  # noqa: E402
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.RandomState(seed)
TOK = {"mean": N.SL_TOK_MEAN, "absmean": N.SL_TOK_ABSMEAN, "max": N.SL_TOK_MAX, "absmax": N.SL_TOK_ABSMAX, "token": N.SL_TOK_TOKEN}


def feq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def plant(x):
    flat = x.view(-1)
    for val in (float("nan"), float("inf"), float("-inf"), -0.0):
        if rng.randint(3) == 0:
            flat[rng.randint(flat.numel())] = val
    if rng.randint(6) == 0 and x.shape[1] > 1:  # a column with +inf and -inf: the sum detector's false positive
        f = rng.randint(x.shape[2])
        x[0, 0, f], x[0, 1, f] = float("inf"), float("-inf")
    return x


for it in range(cases):
    dt = [torch.float32, torch.float16, torch.bfloat16][rng.randint(3)]
    epp = 4 if dt == torch.float32 else 8
    F = epp * int(rng.choice([1, 2, 3, 5, 12, 16, 24, 32, 48, 64, 96, 100, 128, 144, 192, 256, 269, 288, 512, 544]))
    if dt != torch.float32:
        F = min(F, 8 * 288)
    T = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 33, 49, 50, 64, 65, 127, 197, 256, 257, 300, 700]))
    B = int(rng.randint(1, 6)) if F * T > 200000 else int(rng.randint(1, 40))
    agg = ["mean", "absmean", "max", "absmax", "token"][rng.randint(5)]
    pos = int(rng.randint(-T, T))
    print("k2", it, (B, T, F), dt, agg, pos, flush=True)
    x = plant(torch.from_numpy(rng.randn(B, T, F).astype(np.float32)).to(dt))
    xd = x.to(DEV)
    cand = torch.empty((B, F), dtype=torch.bfloat16, device=DEV)
    out = torch.empty((B, F), dtype=torch.float32, device=DEV)
    N.reduce_tokens(xd, TOK[agg], pos, cand, out)
    torch.cuda.synchronize()
    want = oracle.agg_tokens(x.float().numpy(), agg, pos)
    if dt != torch.float32:  # the reference aggregates in the activation dtype: round once
        want = torch.from_numpy(want).to(dt).float().numpy()
    got = out.cpu().numpy()
    if agg in ("mean", "absmean"):
        fin = np.isfinite(want)
        ok = np.array_equal(np.isnan(got), np.isnan(want)) and np.allclose(got[fin], want[fin], rtol=1e-2 if dt != torch.float32 else 2e-5, atol=1e-5)
        ok = ok and np.array_equal(got[np.isinf(want)], want[np.isinf(want)])
    else:
        ok = feq(got, want)
    assert ok, ("mismatch", it, np.abs(np.nan_to_num(got) - np.nan_to_num(want)).max())
    assert feq(cand.float().cpu().numpy(), torch.from_numpy(got).to(torch.bfloat16).float().numpy()), ("cand", it)
    # the same bytes as a channels_last conv map (B, F, h, w): K1's component-contiguous path
    if T > 1 and rng.randint(3) == 0:
        h = [d for d in range(1, T + 1) if T % d == 0][rng.randint(len([d for d in range(1, T + 1) if T % d == 0]))]
        xm = xd.reshape(B, h, T // h, F).permute(0, 3, 1, 2)  # channels_last strides
        cagg = ["max", "mean"][rng.randint(2)]
        N.reduce_conv(xm, N.SL_CONV_MAX if cagg == "max" else N.SL_CONV_MEAN, cand, out)
        torch.cuda.synchronize()
        w2 = oracle.agg_conv(x.float().permute(0, 2, 1).reshape(B, F, h, T // h).contiguous().numpy(), cagg)
        if dt != torch.float32:
            w2 = torch.from_numpy(w2).to(dt).float().numpy()
        g2 = out.cpu().numpy()
        if cagg == "max":
            assert feq(g2, w2), ("conv max", it)
        else:
            fin = np.isfinite(w2)
            assert np.array_equal(np.isnan(g2), np.isnan(w2)) and np.allclose(g2[fin], w2[fin], rtol=1e-2 if dt != torch.float32 else 2e-5, atol=1e-5), ("conv mean", it)
print("fuzz_k2 ok:", cases, "cases, seed", seed, "SL_OPTIONS =", os.environ.get("SL_OPTIONS"))
