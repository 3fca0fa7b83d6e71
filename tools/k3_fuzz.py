"""Fuzz K3 (ActMax.update) against the oracle with a synchronisation after every call; prints the case before it runs."""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oracle
from semanticlens_amd.component_visualization.activation_caching import ActMax

DEV = "cuda:0"
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for it in range(400):
    n, c, k = rng.randint(1, 401), rng.randint(1, 71), rng.randint(1, 41)
    mode = ["aten", "total"][rng.randint(2)]
    acts = (rng.randint(0, 6, size=(n, c)).astype(np.float32) / 4) if rng.randint(2) else rng.randn(n, c).astype(np.float32)
    cuts = sorted(rng.randint(0, n + 1, size=rng.randint(0, 9)).tolist())
    bounds = [0] + cuts + [n]
    print(it, "n", n, "c", c, "k", k, mode, bounds, flush=True)
    am = ActMax(n_collect=k, n_latents=c, tie_mode=mode)
    ref = oracle.ActMaxOracle(k, c, oracle.MODE_ATEN if mode == "aten" else oracle.MODE_TOTAL)
    x = torch.from_numpy(acts).to(DEV)
    for s, e in zip(bounds[:-1], bounds[1:]):
        if e == s:
            continue
        am.update(x[s:e], torch.arange(s, e))
        torch.cuda.synchronize()
        ref.update(acts[s:e], np.arange(s, e))
    v = am.activations.view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(v, ref.vals), "values differ"
    assert np.array_equal(am.sample_ids.numpy(), ref.ids), "ids differ"
print("ok")
