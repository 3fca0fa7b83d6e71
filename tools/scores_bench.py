"""Time the analysis-stage scores (K7/K8/K9) at BASELINE configs[1] scale: layer4 of ResNet-50, k=20, D=512."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import scores  # noqa: E402

DEV = "cuda:0"
torch.manual_seed(0)
V = torch.randn(2048, 20, 512, device=DEV)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"clarity (2048, 20, 512): {timeit(lambda: scores.clarity_score(V)):.3f} ms")
print(f"redundancy (2048, 512): {timeit(lambda: scores.redundancy_score(V.mean(1))):.3f} ms")
print(f"polysemanticity (2048, 20, 512), KMeans(2, n_init=10) per component: {timeit(lambda: scores.polysemanticity_score(V), n=3):.2f} ms")
try:
    import oracle

    t = time.perf_counter()
    oracle.polysemanticity(V[:32].cpu().numpy())
    dt = time.perf_counter() - t
    print(f"scikit-learn on the host (oracle), 32 components: {dt * 1e3:.0f} ms -> {dt / 32 * 2048:.1f} s for 2048")
except Exception as e:
    print("oracle unavailable:", e)
