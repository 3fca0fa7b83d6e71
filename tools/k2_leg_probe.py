"""The tokens_collect leg of bench.py (ViT-B/16 probed model, 12 blocks, K2) with and without the embed on a second stream;
run under `rocprofv3 --kernel-trace` it also yields every K2 dispatch with the kernel in front of it."""
import sys
import types
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import synth  # noqa: E402
from semanticlens_amd.component_visualization import aggregators  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

dev = torch.device("cuda:0")
args = types.SimpleNamespace(k=20, tie_mode="aten", no_self_check=True)
fm = NativeClip(synth.SyntheticClip(device=dev), gemm="bf16x3")
vit = synth.vit_b16().to(dev)
for overlap in ([True, False] if len(sys.argv) < 2 else [sys.argv[1] == "1"]):
    bench.OVERLAP = overlap
    out = bench.collect_leg(dev, fm, args, vit, [f"blocks.{i}" for i in range(12)], aggregators.aggregate_transformer_max,
                            "colreduce", "tokens", steps=8, B=256)
    r = out["roofline"]
    print(f"overlap={overlap}: {out['images_per_s']:.0f} images/s, K2 avg {r['avg_launch_us']:.1f} us = {r['frac']:.3f} ({r['launches']} launches) policy {sorted(set(out.get('reduce_cache_policy', {}).values()))}", flush=True)
