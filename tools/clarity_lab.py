"""K7 at the bench's `scores_full_db` shape (four layers of 192 / 384 / 768 / 1536 components x 20 x 512 in ONE launch, 118 MB) and at
configs[3]'s (12 x 768 x 20 x 1152): kernel time from the dispatch's own timestamps (sl_prof), cold (inputs rotated through > 1 GB) and
warm (the same buffers again: 118 MB sit in the 256 MiB Infinity Cache)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
for tag, Cs, n, D in (("configs[4] db", (192, 384, 768, 1536), 20, 512), ("configs[3] db", (768,) * 12, 20, 1152), ("k=100", (512, 1024, 2048), 100, 512)):
    nbytes = sum(Cs) * n * D * 4
    copies = max(2, (1200 << 20) // nbytes + 1)
    sets = [[torch.randn(c, n, D, device=DEV) for c in Cs] for _ in range(copies)]
    for s in sets[:2]:
        N.clarity_multi(s)
    torch.cuda.synchronize()
    res = {}
    for label, order in (("cold", [i % copies for i in range(3 * copies)]), ("warm", [0] * 12)):
        N.prof_enable(True)
        N.prof_reset()
        for i in order:
            N.clarity_multi(sets[i])
        torch.cuda.synchronize()
        ms, launches, work = N.prof_read(N.SL_PROF_SCORES)
        N.prof_enable(False)
        res[label] = (ms / launches * 1e3, work / ms / 1e6)
    print(f"{tag}: {nbytes / 1e6:.1f} MB per launch | cold {res['cold'][0]:.1f} us = {res['cold'][1]:.0f} GB/s ({res['cold'][1] / 8000:.3f} of 8 TB/s) | "
          f"warm {res['warm'][0]:.1f} us = {res['warm'][1]:.0f} GB/s", flush=True)
    del sets
