"""K1 alone on COLD inputs at the three ResNet-50 layer shapes (B = 256, fp32, and fp16): the launches of `bench.py`'s
`roofline.cold_inputs` leg, in a process of their own so that a rocprofv3 kernel trace of it holds this regime only.
Prints one JSON line with the per-dispatch HIP-event times (`sl_prof`) of the same launches."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
B = 256
out = {}
N.set_reduce_policy(0, 0)  # inputs known to be cold: read-once policy for every byte
for dt, tag in ((torch.float32, "f32"), (torch.float16, "f16")):
    for name, (C, H) in (("layer2", (512, 28)), ("layer3", (1024, 14)), ("layer4", (2048, 7))):
        nbytes = B * C * H * H * torch.finfo(dt).bits // 8
        copies = max(2, (1200 << 20) // nbytes + 1)
        xs = [torch.rand(B, C, H, H, device=DEV).to(dt) for _ in range(copies)]
        cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
        for x in xs:
            N.reduce_conv(x, N.SL_CONV_MAX, cand, None)
        torch.cuda.synchronize()
        N.prof_enable(True)
        N.prof_reset()
        for _ in range(3):
            for x in xs:
                N.reduce_conv(x, N.SL_CONV_MAX, cand, None)
        torch.cuda.synchronize()
        ms, launches, nb = N.prof_read(N.SL_PROF_REDUCE)
        N.prof_enable(False)
        out[f"{tag}_{name}"] = {"bytes_per_launch": nbytes, "launches": launches, "avg_launch_us": ms / launches * 1e3,
                                "GB/s": nb / ms / 1e6, "frac_of_8TBps": nb / ms / 1e6 / 8000.0}
        del xs
print(json.dumps(out))
