"""K2 (token reduce) on the configs[3] block shape (256, 197, 768) fp32 = 155 MB: cold, and right behind producers that
wrote the input microseconds earlier (the residual add of a transformer block; an in-place activation), per dispatch from
sl_prof (HIP events stamped by the dispatch).  K1 on the same bytes viewed as (256, 768, 197) NCHW rows beside it."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
B, T, F = 256, 197, 768
cand = torch.empty(B, F, dtype=torch.bfloat16, device=DEV)
NBUF = 9  # 9 x 155 MB > the Infinity Cache several times over
bufs = [torch.randn(B, T, F, device=DEV) for _ in range(NBUF)]
a, b = torch.randn(B, T, F, device=DEV), torch.randn(B, T, F, device=DEV)
nbytes = B * T * F * 4


def timed(make_input, launch, reps=27):
    for r in range(3):
        launch(make_input(r))
    torch.cuda.synchronize()
    N.prof_enable(True)
    N.prof_reset()
    for r in range(reps):
        launch(make_input(r))
    torch.cuda.synchronize()
    ms, n, _ = N.prof_read(N.SL_PROF_REDUCE)
    N.prof_enable(False)
    us = ms / n * 1e3
    return f"{us:6.1f} us  {nbytes / us / 1e6:5.2f} TB/s = {nbytes / us / 8e6:.3f}"


def k2(x):
    N.reduce_tokens(x, N.SL_TOK_MAX, 0, cand, None)


def k1(x):
    N.reduce_conv(x.view(B, F, T, 1), N.SL_CONV_MAX, cand, None)


def cold(r):
    return bufs[r % NBUF]


def after_add(r):
    x = bufs[r % NBUF]
    torch.add(a, b, out=x)
    return x


def after_relu(r):
    x = bufs[r % NBUF]
    x.relu_()
    return x


def after_add_same(r):
    torch.add(a, b, out=bufs[0])
    return bufs[0]


for name, mk in (("cold (rotating 9 buffers)", cold), ("behind torch.add(a, b, out=x)", after_add), ("behind x.relu_()", after_relu),
                 ("behind torch.add into ONE buffer", after_add_same)):
    print(f"{name:34s} K2 colreduce {timed(mk, k2)}   |  K1 rowreduce on the same bytes {timed(mk, k1)}", flush=True)
