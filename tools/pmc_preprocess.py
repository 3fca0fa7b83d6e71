"""Target for rocprofv3 --pmc passes over K12 (256 images 500x375 -> 224)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

B, h, w, S = 256, 500, 375, 224
px = torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8, device="cuda:0").reshape(-1)
plan, info = N.preprocess_plan([(h, w)] * B, S)
plan_d = plan.to("cuda:0")
for _ in range(4):
    N.preprocess(px, plan_d, info, S, (0.5, 0.5, 0.5), (0.25, 0.25, 0.25))
torch.cuda.synchronize()
