"""K1 on conv maps whose rows are not whole 16-byte pieces (odd H x W: AlexNet 13 x 13 / 27 x 27 / 55 x 55, Inception
17 x 17 / 35 x 35, ...), cold (rotating buffers past the Infinity Cache), fp32 and fp16; per dispatch from sl_prof."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
SHAPES = [(256, 2048, 7, 7), (256, 256, 13, 13), (256, 1024, 13, 13), (256, 768, 17, 17), (256, 384, 27, 27), (128, 288, 35, 35),
          (128, 96, 55, 55), (256, 1024, 14, 14), (64, 64, 111, 111), (256, 197, 1, 768)]
for dt in (torch.float32, torch.float16):
    for shp in SHAPES:
        B, C, H, W = shp
        nbytes = B * C * H * W * (4 if dt == torch.float32 else 2)
        nbuf = max(2, int(1.3e9 // nbytes) + 1)
        bufs = [torch.randn(shp, device=DEV).to(dt) for _ in range(nbuf)]
        cand = torch.empty(B, C, dtype=torch.bfloat16, device=DEV)
        for r in range(3):
            N.reduce_conv(bufs[r % nbuf], N.SL_CONV_MAX, cand, None)
        torch.cuda.synchronize()
        N.prof_enable(True)
        N.prof_reset()
        for r in range(3 * nbuf):
            N.reduce_conv(bufs[r % nbuf], N.SL_CONV_MAX, cand, None)
        torch.cuda.synchronize()
        ms, n, _ = N.prof_read(N.SL_PROF_REDUCE)
        N.prof_enable(False)
        us = ms / n * 1e3
        print(f"{str(dt)[6:]:8s} {str(shp):22s} S = {H * W:6d}  {nbytes / 1e6:7.1f} MB  {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s = {nbytes / us / 8e6:.3f}", flush=True)
        del bufs
