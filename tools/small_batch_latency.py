"""Latency of the native towers at interactive batch sizes: host enqueue time against device time."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth
from semanticlens_amd.foundation_models.native_clip import NativeClip

dev = "cuda:0"
base = synth.SyntheticClip(device=dev)
fm = NativeClip(base)
for B in (1, 8, 64):
    tok = base.tokenize(["a photo of a striped zebra"] * B)
    img = torch.randn(B, 3, 224, 224, device=dev)
    for name, fn in (("encode_text", lambda: fm.encode_text(tok)), ("encode_image", lambda: fm.encode_image(img))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # device-only time: events around a burst
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B:3d} {name:12s}: host enqueue {1e3*(t1-t0)/20:.2f} ms, wall {1e3*(t2-t0)/20:.2f} ms, back-to-back {e0.elapsed_time(e1)/20:.2f} ms per call")
