"""Only the native image tower (ViT-B/32, B=256), for `rocprofv3 --kernel-trace --stats`: per-kernel times of K11."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

base = synth.SyntheticClip(device="cuda:0")
fm = NativeClip(base)
img = torch.randn(256, 3, 224, 224, device="cuda:0")
for _ in range(13):
    fm.encode_image(img)
torch.cuda.synchronize()
