import numpy as np, torch, sys, warnings
sys.path.insert(0, ".")
warnings.filterwarnings("ignore")
import oracle
from semanticlens_amd import scores
DEV = "cuda:0"
cases = [(256, 20, 512, "random"), (256, 20, 512, "blobs"), (128, 10, 64, "random"), (64, 33, 100, "blobs"), (64, 100, 32, "random"), (96, 20, 1152, "blobs3")]
for C, n, D, kind in cases:
    rng = np.random.RandomState(C + n + D)
    V = rng.randn(C, n, D).astype(np.float32)
    if kind.startswith("blobs"):
        nb = 3 if kind == "blobs3" else 2
        centers = rng.randn(C, nb, D).astype(np.float32) * 2
        assign = rng.randint(0, nb, size=(C, n))
        V = centers[np.arange(C)[:, None], assign] + 0.5 * V
    want = oracle.polysemanticity(V)
    got = scores.polysemanticity_score(torch.from_numpy(V).to(DEV)).cpu().numpy()
    d = np.abs(got - want)
    print(C, n, D, kind, "agree1e-5", (d <= 1e-5).mean(), "agree1e-9", (d <= 1e-9).mean(), "max", d.max(), "bad idx", np.nonzero(d > 1e-5)[0][:10], flush=True)
rng = np.random.RandomState(4)
V = rng.randn(1536, 20, 512).astype(np.float32)
V[::7, :9] += 2.0 * rng.randn(220, 1, 512).astype(np.float32)
got = scores.polysemanticity_score(torch.from_numpy(V).to(DEV)).cpu().numpy()
sub = np.arange(0, 1536, 6)
want = oracle.polysemanticity(V[sub])
d = np.abs(got[sub] - want)
print("config4", (d <= 1e-5).mean(), (d <= 1e-9).mean(), d.max(), np.nonzero(d > 1e-5)[0][:10])
