"""NativeSigLip at the SigLIP-so400m geometry (27 x 1152, 16 heads of 72, MLP 4304, patch 14, 256 tokens at 224 px;
random init) against the wrapped transformers modules: image B = 64, text B = 1024 x ctx 64."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth  # noqa: E402
from semanticlens_amd.foundation_models import NativeSigLip  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


base = synth.SyntheticSigLip(device=DEV)
fm = NativeSigLip(base)
img = torch.randn(64, 3, 224, 224, device=DEV)
tok = base.tokenize(["a photo of a striped zebra in the tall grass near the river"] * 1024)
a, b = fm.encode_image(img[:8]), base.encode_image(img[:8])
print("image features: max relative difference", ((a - b).abs().max() / b.abs().max()).item())
a, b = fm.encode_text(tok[:8]), base.encode_text(tok[:8])
print("text features: max relative difference", ((a - b).abs().max() / b.abs().max()).item())
print(f"image B=64: native {timeit(lambda: fm.encode_image(img)):.1f} ms, torch {timeit(lambda: base.encode_image(img)):.1f} ms")
print(f"text B=1024: native {timeit(lambda: fm.encode_text(tok)):.1f} ms, torch {timeit(lambda: base.encode_text(tok)):.1f} ms")
