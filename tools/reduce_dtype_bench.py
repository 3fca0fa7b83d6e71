"""K1/K2 bandwidth by activation dtype and layout (cold inputs rotated through > 1 GB)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

dev = "cuda:0"
N.set_reduce_policy(0, 0)


def bench(shape, dtype, cl, agg, reps=8):
    nbuf = max(2, int(1.2e9 // (torch.tensor(shape).prod().item() * torch.finfo(dtype).bits // 8)))
    xs = []
    for _ in range(nbuf):
        x = torch.randn(shape, device=dev, dtype=torch.float32).to(dtype)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last)
        xs.append(x)
    cand = torch.empty(shape[:2], dtype=torch.bfloat16, device=dev)
    for x in xs[:2]:
        N.reduce_conv(x, agg, cand, None)
    torch.cuda.synchronize()
    N.prof_enable(True); N.prof_reset()
    for i in range(reps):
        N.reduce_conv(xs[i % nbuf], agg, cand, None)
    ms, n, nb = N.prof_read(N.SL_PROF_REDUCE)
    N.prof_enable(False)
    return nb / ms / 1e6, ms / n * 1e3


for shape in ((256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7)):
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        for cl in (False, True):
            for agg, an in ((N.SL_CONV_MAX, "max"), (N.SL_CONV_MEAN, "mean")):
                gbps, us = bench(shape, dtype, cl, agg)
                print(f"{str(shape):22s} {str(dtype):15s} {'channels_last' if cl else 'nchw':13s} {an:4s} {gbps:7.0f} GB/s {us:8.1f} us", flush=True)


def bench_tokens(shape, dtype, agg, reps=8):
    nbuf = max(2, int(1.2e9 // (shape[0] * shape[1] * shape[2] * torch.finfo(dtype).bits // 8)))
    xs = [torch.randn(shape, device=dev, dtype=torch.float32).to(dtype) for _ in range(nbuf)]
    cand = torch.empty((shape[0], shape[2]), dtype=torch.bfloat16, device=dev)
    for x in xs[:2]:
        N.reduce_tokens(x, agg, 0, cand, None)
    torch.cuda.synchronize()
    N.prof_enable(True); N.prof_reset()
    for i in range(reps):
        N.reduce_tokens(xs[i % nbuf], agg, 0, cand, None)
    ms, n, nb = N.prof_read(N.SL_PROF_REDUCE)
    N.prof_enable(False)
    return nb / ms / 1e6, ms / n * 1e3


for shape in ((256, 197, 768), (256, 50, 768), (48, 729, 1152)):
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        for agg, an in ((N.SL_TOK_MEAN, "mean"), (N.SL_TOK_MAX, "max")):
            gbps, us = bench_tokens(shape, dtype, agg)
            print(f"tokens {str(shape):18s} {str(dtype):15s} {an:4s} {gbps:7.0f} GB/s {us:8.1f} us", flush=True)
