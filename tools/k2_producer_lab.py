"""K1/K2 on ConvNeXt-L's stage-0 output shape behind the producers a real block ends with: what the cache policy can and cannot do.
Producers: `inplace` (x += c: two streams, what tools/k2_lab.py's "pipe" regime uses), `add3` (out = x + h: three streams through
the 256 MiB Infinity Cache, what ConvNeXt's residual add does), `add3+gemm` (the same, then an unrelated GEMM before the reduce).
Policies: (nt_min_bytes, tail_bytes) of sl_set_reduce_policy."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

dev = "cuda:0"
MiB = 1 << 20
for shape in ((256, 3136, 192), (256, 784, 384), (256, 197, 768)):
    nb = shape[0] * shape[1] * shape[2] * 4
    xs = [torch.randn(shape, device=dev) for _ in range(3)]
    hs = [torch.randn(shape, device=dev) for _ in range(3)]
    outs = [torch.empty(shape, device=dev) for _ in range(3)]
    cand = torch.empty((shape[0], shape[2]), dtype=torch.bfloat16, device=dev)
    a, b = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
    for prod in ("cold", "inplace", "add3", "add3+gemm"):
        row = [f"{str(shape):16s} {nb / 1e6:5.0f} MB  {prod:10s}"]
        for pol in ((0, 0), (0, 48 * MiB), (0, 80 * MiB), (0, 128 * MiB), (0, 240 * MiB), (1 << 60, 0)):
            N.set_reduce_policy(*pol)
            N.prof_enable(True)
            N.prof_reset()
            for i in range(9):
                j = i % 3
                if prod == "inplace":
                    torch.add(outs[j], 0.5, out=outs[j])
                elif prod.startswith("add3"):
                    torch.add(xs[j], hs[j], out=outs[j])
                    if prod.endswith("gemm"):
                        torch.mm(a, b)
                N.reduce_tokens(outs[j], N.SL_TOK_MAX, 0, cand, None)
            torch.cuda.synchronize()
            ms, n, nbytes = N.prof_read(N.SL_PROF_REDUCE)
            N.prof_enable(False)
            name = "all-nt" if pol == (0, 0) else ("no-nt" if pol[0] > 1 << 50 else f"tail{pol[1] // MiB}")
            row.append(f"{name} {nbytes / ms / 1e6 / 8000:.3f}")
        print("  ".join(row), flush=True)
    del xs, hs, outs
N.set_reduce_policy(None, None)
