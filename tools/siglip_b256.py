"""so400m image tower at B = 256 (what the embed stage and bench.py's configs[3] leg use): wall time per encode."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth  # noqa: E402
from semanticlens_amd.foundation_models import NativeSigLip  # noqa: E402

fm = NativeSigLip(synth.SyntheticSigLip(device="cuda:0"))
for B in (64, 256):
    x = torch.randn(B, 3, 224, 224, device="cuda:0")
    for _ in range(2):
        fm.encode_image(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        fm.encode_image(x)
    torch.cuda.synchronize()
    print(f"so400m image B={B}: {(time.perf_counter() - t) / 5 * 1e3:.1f} ms")
