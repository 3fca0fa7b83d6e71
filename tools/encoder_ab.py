"""A/B of library builds on the native ViT-B/32 image tower (set SEMANTICLENS_AMD_LIB): wall per encode at B = 256."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

fm = NativeClip(synth.SyntheticClip(device="cuda:0"))
img = torch.randn(256, 3, 224, 224, device="cuda:0")
for _ in range(5):
    fm.encode_image(img)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        fm.encode_image(img)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t) / 20 * 1e3)
print(f"image B=256: {best:.3f} ms per encode (best of 5 x 20)")
