"""Interactive probing latency: Lens.text_probing of 1 / 8 prompts (with and without templates) and Lens.image_probing of
one image against a 12 x 768 x 512 concept DB, native towers vs the wrapped torch model; ms per call, synchronised."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth  # noqa: E402
from semanticlens_amd import Lens  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

dev = torch.device("cuda:0")
base = synth.SyntheticClip(device=dev)
g = torch.Generator(device=dev).manual_seed(3)
db = {f"block{i}": torch.randn(768, 512, device=dev, generator=g) for i in range(12)}
templates = [f"a photo of a {{}} {i}" for i in range(16)]
img = synth.synth_images_u8(torch.arange(1, device=dev))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, fm in (("native", NativeClip(base)), ("torch ", base)):
    lens = Lens(fm, device=dev)
    x = fm.preprocess(img) if hasattr(fm, "preprocess") else img
    print(f"{name}: text_probing 1 prompt {timed(lambda: lens.text_probing(['zebra'], db)):.2f} ms, "
          f"8 prompts {timed(lambda: lens.text_probing(['zebra', 'dog', 'cat', 'sky', 'wheel', 'grass', 'red', 'wood'], db)):.2f} ms, "
          f"8 prompts x 16 templates {timed(lambda: lens.text_probing(['zebra', 'dog', 'cat', 'sky', 'wheel', 'grass', 'red', 'wood'], db, templates=templates)):.2f} ms, "
          f"image_probing 1 image {timed(lambda: lens.image_probing(x, db)):.2f} ms", flush=True)
