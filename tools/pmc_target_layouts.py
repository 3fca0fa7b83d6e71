"""Target for `rocprofv3 --pmc FETCH_SIZE`: K1 on the other layouts / dtypes at the ResNet-50 shapes (B = 256), cold
inputs (rotation over > 256 MiB): channels_last fp32, NCHW and channels_last fp16 / bf16.  One kernel name per case."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
N.set_reduce_policy(0, 0)
for (B, C, H, W) in ((256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7)):
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        for cl in (False, True):
            if dt == torch.float32 and not cl:
                continue  # the headline kernels: tools/pmc_target.py
            nbuf = 6 if dt != torch.float32 else 4
            xs = [torch.randn(B, C, H, W, device=DEV).to(dt) for _ in range(nbuf)]
            if cl:
                xs = [x.contiguous(memory_format=torch.channels_last) for x in xs]
            cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
            for i in range(nbuf):
                N.reduce_conv(xs[i], N.SL_CONV_MAX, cand, None)
            torch.cuda.synchronize()
            del xs
