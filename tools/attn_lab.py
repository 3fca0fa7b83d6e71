"""sl_attention (fp32 MFMA) vs sl_attention_bf16x3 over batch sizes: how long one workgroup's life is (one round of
workgroups fits the chip up to B * H = 1024) against the steady-state rate."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def wall(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


for T, H, D in ((50, 12, 64), (197, 12, 64), (256, 16, 72), (77, 8, 64)):
    for B in (16, 64, 256) if T < 100 else (16, 64):
        qkv = torch.randn(B * T, 3 * H * D, device=DEV)
        sp = N.Split(B * T, H * D, DEV)
        a = wall(lambda: N.attention(qkv, B, T, H, D, False, out_split=sp))
        b = wall(lambda: N.attention(qkv, B, T, H, D, False, out_split=sp, bf16x3=True))
        mb = (qkv.numel() * 4 + B * T * H * D * 4) / 1e6
        print(f"T={T:4d} H={H:2d} D={D:3d} B={B:4d} ({B * H:5d} workgroups, {mb:6.1f} MB): fp32 MFMA {a:7.1f} us   bf16x3 {b:7.1f} us", flush=True)
