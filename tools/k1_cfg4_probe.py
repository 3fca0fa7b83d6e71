"""The bench's configs[4] collect leg alone (ConvNeXt-L, 4 stage outputs, channels_last -> colreduce2), for a kernel trace: which
of the four launches costs what inside the pipeline.  python tools/k1_cfg4_probe.py [overlap 0|1]"""
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
model = synth.convnext_l().to(dev)
bench.OVERLAP = len(sys.argv) > 1 and sys.argv[1] == "1"
out = bench.collect_leg(dev, fm, args, model, [f"stages.{i}" for i in range(4)], aggregators.aggregate_conv_max, "colreduce2", "cfg4", steps=6, B=256)
r = out["roofline"]
print(f"overlap={bench.OVERLAP}: {out['images_per_s']:.0f} images/s, K1 avg {r['avg_launch_us']:.1f} us = {r['frac']:.3f} ({r['launches']} launches) policy {sorted(set(out.get('reduce_cache_policy', {}).values()))}", flush=True)
