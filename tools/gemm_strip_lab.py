"""What a column-strip split of the N = 1152 GEMMs (so400m o-proj / fc2: 4.5 column tiles of 256) could buy: times of the full GEMM,
of its first 1024 columns and of the 128-column remainder, per tile variant (option g3_tile, here through SL_OPTIONS: run once per value).
    for t in 0 8 256 128 1280 160; do SL_OPTIONS=g3_tile=$t python tools/gemm_strip_lab.py; done"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    N.prof_enable(True)
    N.prof_reset()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, n, _ = N.prof_read(N.SL_PROF_GEMM)
    N.prof_enable(False)
    return ms / max(n, 1) * 1e3


tile = os.environ.get("SL_OPTIONS", "auto")
for M in (16384, 65536):
    for K in (1152, 4304):
        x = N.Split.of(torch.randn(M, K, device=DEV, generator=g))
        row = [f"tile={tile:>5s} M={M:6d} K={K:5d}:"]
        for Nn in (1152, 1024, 128, 3456, 3328):
            w = N.Split.of(torch.randn(Nn, K, device=DEV, generator=g) * 0.03)
            b = torch.randn(Nn, device=DEV, generator=g)
            res = torch.randn(M, Nn, device=DEV, generator=g)
            try:
                us = timed(lambda: N.linear3(x, w, b, residual=res, out=res))
                row.append(f"N={Nn}: {us:7.1f} us")
            except Exception as e:  # a forced variant may refuse a shape
                row.append(f"N={Nn}: n/a ({type(e).__name__})")
            del w, b, res
        print("  ".join(row), flush=True)
        del x
