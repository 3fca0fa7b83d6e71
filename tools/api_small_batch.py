"""The API path (Lens.compute_concept_db over host Datasets) at the reference's default batch_size = 32, with the embed
stage encoding every loader batch on its own (embed_accumulate = 0) and holding batches back until 256 images are there."""
import sys
import types
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import synth  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

dev = torch.device("cuda:0")
model = synth.resnet50().to(dev).eval()
base = synth.SyntheticClip(device=dev)
for B in (32, 64, 256):
    for acc in (0, 256):
        NativeClip.embed_accumulate = acc
        args = types.SimpleNamespace(batch=B, k=20, api_images=4096)
        out = bench.api_path_leg(dev, model, base, args)
        print(f"batch_size {B:3d}, embed_accumulate {acc:3d}: single pass {out['api_path_images_per_s']:6.0f} images/s, two passes {out['two_pass_images_per_s']:6.0f}", flush=True)
