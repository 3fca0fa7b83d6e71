"""Host time to ENQUEUE one bench step (no synchronisation) against its device time: how close the timed loop is to
being launch-bound.  python tools/host_overhead.py  (GPU)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import synth  # noqa: E402
from semanticlens_amd import _native as N  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

dev = torch.device("cuda:0")
model = synth.resnet50().to(dev)
fm = NativeClip(synth.SyntheticClip(device=dev))
B, K = 256, 12
batches = [synth.synth_images_u8(torch.arange(s * B, (s + 1) * B, device=dev)) for s in range(K)]
cv = bench.make_cv(model, K * B, 20, "total")
bench.run_steps(cv, fm, batches[:3], 0, 3 * B)
torch.cuda.synchronize()
for prof in (False, True):
    N.prof_enable(prof)
    N.prof_reset()
    cv = bench.make_cv(model, K * B, 20, "total")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_steps(cv, fm, batches, 0, K * B)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"sl_prof {'on ' if prof else 'off'}: host enqueue {1e3 * (t1 - t0) / K:.1f} ms/step, wall {1e3 * (t2 - t0) / K:.1f} ms/step")
N.prof_enable(False)
# the two halves separately
with torch.no_grad():
    x = synth.normalize_u8(batches[0], synth.IMAGENET_MEAN, synth.IMAGENET_STD)
    for name, fn in (("model forward (no hooks)", lambda: model(x)), ("native encode_image", lambda: fm.encode_image(fm.preprocess(batches[0])))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: host enqueue {1e2 * (t1 - t0):.1f} ms, wall {1e2 * (t2 - t0):.1f} ms per call")
