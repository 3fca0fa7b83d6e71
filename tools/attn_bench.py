"""Time sl_attention at the CLIP tower shapes (the VALU kernel of round 1 left the library in round 6: tools/native/attention_valu_lab.hpp)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

for B, T, H, causal in ((256, 50, 12, False), (1024, 77, 8, True), (64, 197, 12, False)):
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda:0")
    out = torch.empty(B * T, H * 64, device="cuda:0")
    for _ in range(3):
        N.attention(qkv, B, T, H, 64, causal, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        N.attention(qkv, B, T, H, 64, causal, out=out)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    flops = 4.0 * B * H * T * T * 64 * (0.5 if causal else 1.0)
    print(f"B={B} T={T} H={H} causal={causal}: {ms * 1e3:.1f} us, {flops / ms / 1e9:.1f} TFLOP/s, {(qkv.numel() + out.numel()) * 4 / ms / 1e6:.0f} GB/s")
