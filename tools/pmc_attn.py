"""Target for rocprofv3 --pmc passes over `attention_bf16x3_kernel` at the so400m shape (B = 256, T = 256, H = 16, head_dim 72),
(profiles/r06_attention_split_ab.txt holds the round-6 counters of the shipped launch shape and of the reverted experiment)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

B, T, H, D = 256, 256, 16, 72
qkv = torch.randn(B * T, 3 * H * D, device="cuda:0")
out = torch.empty(B * T, H * D, device="cuda:0")
for _ in range(4):
    N.attention(qkv, B, T, H, D, False, out=out, bf16x3=True)
torch.cuda.synchronize()
