"""K3 in the reference's tie order (`SL_TIES_ATEN`): fuzz against the oracle at the shapes the wave-per-row kernel branches on,
then time `sl_actmax_update` per launch (HIP events around 200 launches on one stream).

    python tools/k3_bench.py            # fuzz + timing with the default implementation (wave per row)
    (SL_K3_ATEN_IMPL=lane, the round-4 kernel as an A/B, was a switch of rounds 4-5: profiles/r05_k3_lane.txt)
"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oracle  # the checker
from semanticlens_amd import _native as N
from semanticlens_amd.component_visualization.activation_caching import ActMax

DEV = torch.device("cuda:0")


def make_acts(rng, kind, n, c):
    if kind == 0:
        a = rng.standard_normal((n, c)).astype(np.float32)
    elif kind == 1:  # heavy ties
        a = rng.integers(0, 6, size=(n, c)).astype(np.float32) / 4
    elif kind == 2:  # all equal, signed zeros
        a = np.zeros((n, c), np.float32)
        a[::3] = -0.0
    elif kind == 3:  # ascending / descending streams (median-of-3 stress)
        a = np.arange(n, dtype=np.float32)[:, None] * np.where(np.arange(c) % 2 == 0, 1.0, -1.0)[None, :].astype(np.float32)
    elif kind == 4:  # post-ReLU: half the entries exactly 0, the rest bf16-coarse
        a = np.maximum(rng.standard_normal((n, c)), 0).astype(np.float32)
    else:  # NaN / inf sprinkled over ties
        a = rng.integers(0, 50, size=(n, c)).astype(np.float32) / 8 - 1
        a[rng.random((n, c)) < 0.05] = np.nan
        a[rng.random((n, c)) < 0.02] = np.inf
        a[rng.random((n, c)) < 0.02] = -np.inf
    return a


def fuzz(seed=0, iters=120):
    rng = np.random.default_rng(seed)
    shapes = [  # (n samples, batch, C, k): the wave kernel's branches
        (1024, 256, 2048, 20), (512, 256, 512, 100), (300, 64, 37, 20), (257, 256, 8, 1), (700, 100, 130, 17), (2600, 1300, 24, 20),
        (900, 300, 1030, 33), (64, 7, 5, 40), (640, 320, 260, 64), (2200, 1100, 9, 3), (128, 32, 1, 20), (520, 130, 2, 65),
    ]
    t0 = time.time()
    for it in range(iters):
        n, B, c, k = shapes[it % len(shapes)]
        kind = int(rng.integers(0, 6))
        acts = make_acts(rng, kind, n, c)
        am = ActMax(n_collect=k, n_latents=c, tie_mode="aten")
        ref = oracle.ActMaxOracle(k, c, oracle.MODE_ATEN)
        x = torch.from_numpy(acts).to(DEV)
        for s in range(0, n, B):
            e = min(n, s + B)
            am.update(x[s:e], torch.arange(s, e))
            ref.update(acts[s:e], np.arange(s, e))
            v = am.activations.view(torch.int16).numpy().view(np.uint16)
            assert np.array_equal(v, ref.vals), ("values differ", it, n, B, c, k, kind, s)
            assert np.array_equal(am.sample_ids.numpy(), ref.ids), ("ids differ", it, n, B, c, k, kind, s)
    print(f"fuzz ok: {iters} streams checked after every batch against the oracle in {time.time() - t0:.1f} s", flush=True)


def timing():
    impl = os.environ.get("SL_K3_ATEN_IMPL", "wave")
    g = torch.Generator(device="cpu").manual_seed(0)
    print(f"impl={impl}: sl_actmax_update(SL_TIES_ATEN), avg us per launch over 200 launches (steady state: a full top-k state, post-ReLU bf16 candidates)")
    for B in (256, 64):
        for k in (20, 100):
            for C in (512, 1024, 2048, 9216):
                vals = torch.empty((C, k), dtype=torch.bfloat16, device=DEV)
                ids = torch.empty((C, k), dtype=torch.int64, device=DEV)
                N.actmax_init(vals, ids)
                ws = torch.empty(N.actmax_aten_ws_bytes(C, k, B), dtype=torch.uint8, device=DEV)
                cands = [(torch.randn(B, C, generator=g).relu() * (1 + 0.1 * i)).to(torch.bfloat16).to(DEV) for i in range(8)]
                for i in range(16):  # fill the state, warm up
                    N.actmax_update(vals, ids, cands[i % 8], None, i * B, B, N.SL_TIES_ATEN, ws)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 200
                e0.record()
                for i in range(reps):
                    N.actmax_update(vals, ids, cands[i % 8], None, (16 + i) * B, B, N.SL_TIES_ATEN, ws)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1000 / reps
                print(f"  B={B:4d} k={k:3d} C={C:5d}: {us:8.1f} us per launch  ({B / us:.2f} M images/s per layer)", flush=True)


if __name__ == "__main__":
    if "--no-fuzz" not in sys.argv:
        fuzz()
    timing()
