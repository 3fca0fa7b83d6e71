"""Time the native CLIP towers (K11) against the wrapped torch modules: ViT-B/32 image B=256, text B=1024."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


base = synth.SyntheticClip(device=DEV)
fm = NativeClip(base)
img = torch.randn(256, 3, 224, 224, device=DEV)
tok = base.tokenize(["a photo of a striped zebra in the grass"] * 1024)
print(f"image B=256: native {timeit(lambda: fm.encode_image(img)):.2f} ms, torch {timeit(lambda: base.encode_image(img)):.2f} ms")
print(f"text B=1024: native {timeit(lambda: fm.encode_text(tok)):.2f} ms, torch {timeit(lambda: base.encode_text(tok)):.2f} ms")
