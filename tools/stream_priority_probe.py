"""The bench step (ResNet-50 forward + collect on one stream, native ViT-B/32 embed on a second) with stream priorities:
does a high-priority forward stream, or a high-priority embed stream, change the step?"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import synth  # noqa: E402
from semanticlens_amd import _native as N  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

DEV = torch.device("cuda:0")
B, nb, k = 256, 24, 20
model = synth.resnet50().to(DEV).eval()
fm = NativeClip(synth.SyntheticClip(device=DEV), gemm="bf16x3")
pool = [synth.synth_images_u8(torch.arange(s * B, (s + 1) * B, device=DEV)) for s in range(6)]
batches = [pool[s % 6] for s in range(nb)]
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
for name, pm, ps in (("main 0 / side 0", 0, 0), ("main -1 / side 0", -1, 0), ("main 0 / side -1", 0, -1), ("main 0 / side 0", 0, 0),
                     ("main -1 / side 0", -1, 0), ("main 0 / side -1", 0, -1)):
    main_s, side_s = torch.cuda.Stream(priority=pm), torch.cuda.Stream(priority=ps)
    best = None
    for rep in range(4):
        cv = bench.make_cv(model, nb * B, k, "aten")
        N.prof_enable(True)
        N.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(main_s):
            mk = torch.cuda.Stream
            torch.cuda.Stream = lambda *a, **kw: side_s  # run_steps creates its side stream: hand it ours
            try:
                emb = bench.run_steps(cv, fm, batches, 0, nb * B)
            finally:
                torch.cuda.Stream = mk
            bench.finish_job(cv, emb, 0, nb * B, False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        red_ms, red_n, red_bytes = N.prof_read(N.SL_PROF_REDUCE)
        N.prof_enable(False)
        if rep and (best is None or dt < best[0]):
            best = (dt, red_bytes / red_ms / 1e9 / 8.0)
    print(f"{name:18s}: {nb * B / best[0]:7.0f} images/s  {best[0] / nb * 1e3:6.2f} ms/batch   K1 {best[1]:.3f} of 8 TB/s", flush=True)
