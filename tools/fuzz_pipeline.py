"""API-level fuzz: Lens.compute_concept_db over tiny / ragged datasets in every mode combination, against the oracle fed
CPU activations of the same integer-valued model (exact in any summation order).  python tools/fuzz_pipeline.py [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle  # noqa: E402
from helpers import FakeVLM, TensorPairDataset, make_int_conv_model, make_int_images  # noqa: E402
from semanticlens_amd import Lens  # noqa: E402
from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators  # noqa: E402

DEV = "cuda:0"
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
AGGS = {"max": aggregators.aggregate_conv_max, "mean": aggregators.aggregate_conv_mean}
for it in range(60):
    n = int(rng.choice([1, 2, 3, 5, 17, 33, 70]))
    bs = int(rng.choice([1, 2, 4, 16, 64]))
    k = int(rng.choice([1, 2, 5, 9, 50]))
    mode = ["aten", "total"][rng.randint(2)]
    single, refonly, prefetch = bool(rng.randint(2)), bool(rng.randint(2)), bool(rng.randint(2))
    agg = ["max", "mean"][rng.randint(2)]
    print(it, "n", n, "bs", bs, "k", k, mode, "single" if single else "two-pass", "refonly" if refonly else "", "prefetch" if prefetch else "sync", agg, flush=True)
    x = make_int_images(n, seed=int(rng.randint(1000)))
    cpu_model = make_int_conv_model()
    ds = TensorPairDataset(x, name=f"fz{it}")
    cv = ActivationComponentVisualizer(make_int_conv_model().to(DEV), ds, ds, ["0", "2"], num_samples=k,
                                       aggregate_fn=AGGS[agg], cache_dir=None, tie_mode=mode)
    cv.prefetch = prefetch
    fm = FakeVLM().to(DEV)
    db = Lens(fm, device=DEV).compute_concept_db(cv, batch_size=bs, single_pass=single, referenced_only=refonly)
    torch.cuda.synchronize()
    # oracle: CPU activations of the same model, batch by batch like the visualizer
    feats = {}
    hooks = [cpu_model[int(nm)].register_forward_hook(lambda m, i, o, nm=nm: feats.__setitem__(nm, o.detach().numpy())) for nm in ("0", "2")]
    refs = {nm: oracle.ActMaxOracle(k, c, oracle.MODE_ATEN if mode == "aten" else oracle.MODE_TOTAL) for nm, c in (("0", 8), ("2", 16))}
    with torch.no_grad():
        for s in range(0, n, bs):
            cpu_model(x[s:s + bs])
            for nm in refs:
                refs[nm].update(oracle.agg_conv(feats[nm], agg), np.arange(s, min(n, s + bs)))
    for h in hooks:
        h.remove()
    emb = FakeVLM().encode_image(x).numpy()
    for nm in refs:
        am = cv.actmax_cache.cache[nm]
        v = am.activations.view(torch.int16).numpy().view(np.uint16)
        assert np.array_equal(v, refs[nm].vals), (nm, "values")
        assert np.array_equal(am.sample_ids.numpy(), refs[nm].ids), (nm, "ids")
        assert np.array_equal(db[nm].cpu().numpy(), oracle.gather_rows(emb, refs[nm].ids)), (nm, "concept_db")
print("ok")
