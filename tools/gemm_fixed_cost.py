"""What a GEMM launch costs besides its k loop: (12 800 x 768) outputs at K = 32 .. 3072, bare epilogue (fp32 stores), per
dispatch from sl_prof (HIP events stamped by the dispatch itself).  SL_OPTIONS=g3_tile=<n> selects the kernel."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

M, W = 12800, 768
out = torch.empty(M, W, device="cuda:0")
for K in (32, 64, 128, 256, 768, 1536, 3072):
    a, b = N.Split.of(torch.randn(M, K, device="cuda:0")), N.Split.of(torch.randn(W, K, device="cuda:0"))
    for _ in range(3):
        N.linear3(a, b, out=out)
    torch.cuda.synchronize()
    N.prof_enable(True)
    N.prof_reset()
    for _ in range(20):
        N.linear3(a, b, out=out)
    torch.cuda.synchronize()
    ms, n, _ = N.prof_read(N.SL_PROF_GEMM)
    N.prof_enable(False)
    print(f"K = {K:5d}: {ms / n * 1e3:7.1f} us   ({K // 32} k-tiles)")
