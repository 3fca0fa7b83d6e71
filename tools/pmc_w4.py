"""Target for rocprofv3 passes over one encoder GEMM (fc2 of ViT-B/32 at B = 256: 12 800 x 768 x 3072), bare epilogue."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

M, W, F = 12800, 768, 3072
hid = torch.randn(M, F, device="cuda:0")
w = torch.randn(W, F, device="cuda:0") * 0.02
sh, sw = N.Split.of(hid), N.Split.of(w)
out = torch.empty(M, W, device="cuda:0")
for _ in range(12):
    N.linear3(sh, sw, out=out)
torch.cuda.synchronize()
