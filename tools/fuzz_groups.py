"""API-level fuzz of the layer groups: Lens.compute_concept_db on models with REPEATED identical blocks (tokens, channels_last and
NCHW convolutions), random subsets of hooked layers, tiny / ragged datasets, against the oracle fed the device's own activations
(taps registered in front of the visualizer's hooks).  python tools/fuzz_groups.py [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch
from torch import nn

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle  # noqa: E402
from helpers import FakeVLM, TensorPairDataset, make_int_images  # noqa: E402
from semanticlens_amd import Lens  # noqa: E402
from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators  # noqa: E402

DEV = "cuda:0"
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)


class Tokens(nn.Module):
    def __init__(self, depth, width):
        super().__init__()
        self.inp = nn.Linear(3 * 16, width)
        self.blocks = nn.ModuleList([nn.Sequential(nn.Linear(width, width), nn.Tanh()) for _ in range(depth)])
        self.head = nn.Linear(width, 5)

    def forward(self, x):  # (B, 3, 16, 16) -> 16 tokens of 48 features
        x = self.inp(x.permute(0, 2, 1, 3).reshape(x.shape[0], 16, 48))
        for b in self.blocks:
            x = b(x)
        return self.head(x.mean(1))


class Convs(nn.Module):
    def __init__(self, depth, width):
        super().__init__()
        self.inp = nn.Conv2d(3, width, 3, padding=1)
        self.blocks = nn.ModuleList([nn.Sequential(nn.Conv2d(width, width, 3, padding=1), nn.ReLU()) for _ in range(depth)])
        self.down = nn.Sequential(nn.Conv2d(width, width, 2, 2), nn.ReLU())  # another shape
        self.head = nn.Linear(width, 5)

    def forward(self, x):
        x = self.inp(x)
        for b in self.blocks:
            x = b(x)
        return self.head(self.down(x).mean((2, 3)))


for it in range(40):
    kind = ["tokens", "conv_cl", "conv_nchw"][rng.randint(3)]
    depth, width = int(rng.randint(2, 6)), int(rng.choice([8, 16, 24, 64]))
    n = int(rng.choice([1, 3, 17, 33, 70]))
    bs = int(rng.choice([1, 4, 16, 64]))
    k = int(rng.choice([1, 3, 9, 50]))
    single = bool(rng.randint(2))
    torch.manual_seed(int(rng.randint(10**6)))
    if kind == "tokens":
        model, agg, oagg = Tokens(depth, width), aggregators.aggregate_transformer_max, lambda a: oracle.agg_tokens(a, "max")
    else:
        model, agg, oagg = Convs(depth, width), aggregators.aggregate_conv_max, lambda a: oracle.agg_conv(a, "max")
    model = model.to(DEV).eval()
    if kind == "conv_cl":
        model = model.to(memory_format=torch.channels_last)
    names = [f"blocks.{i}" for i in range(depth)] + (["down"] if kind != "tokens" else [])
    hooked = [nm for nm in names if rng.rand() < 0.8] or names[:1]
    print(it, kind, "depth", depth, "width", width, "n", n, "bs", bs, "k", k, "single" if single else "two-pass", hooked, flush=True)
    x = make_int_images(n, seed=int(rng.randint(1000)))
    ds = TensorPairDataset(x, name=f"fg{it}")
    mods = dict(model.named_modules())
    taps = {nm: [] for nm in hooked}
    handles = [mods[nm].register_forward_hook(lambda m, i, o, nm=nm: taps[nm].append(o.detach().float().cpu().numpy())) for nm in hooked]
    cv = ActivationComponentVisualizer(model, ds, ds, hooked, num_samples=k, aggregate_fn=agg, cache_dir=None, tie_mode="aten")
    fm = FakeVLM().to(DEV)
    db = Lens(fm, device=DEV).compute_concept_db(cv, batch_size=bs, single_pass=single)
    torch.cuda.synchronize()
    for h in handles:
        h.remove()
    emb = FakeVLM().encode_image(x).numpy()
    for nm in hooked:
        assert len(taps[nm]) == -(-n // bs), (nm, len(taps[nm]))
        ref, start = None, 0
        for a in taps[nm]:
            red = oagg(a)
            ref = ref or oracle.ActMaxOracle(k, red.shape[1], oracle.MODE_ATEN)
            ref.update(red, np.arange(start, start + a.shape[0]))
            start += a.shape[0]
        am = cv.actmax_cache.cache[nm]
        assert np.array_equal(am.activations.view(torch.int16).numpy().view(np.uint16), ref.vals), (nm, "values")
        assert np.array_equal(am.sample_ids.numpy(), ref.ids), (nm, "ids")
        assert np.array_equal(db[nm].cpu().numpy(), oracle.gather_rows(emb, ref.ids)), (nm, "concept_db")
    print("   groups:", [g["layers"] for g in cv.actmax_cache._groups], flush=True)
print("ok")
