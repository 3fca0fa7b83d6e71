"""Does running the image tower as two half-batches on two HIP streams fill the chip better than one batch?"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth
from semanticlens_amd.foundation_models.native_clip import NativeClip

dev = "cuda:0"
fm = NativeClip(synth.SyntheticClip(device=dev))
img = torch.randn(256, 3, 224, 224, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def whole():
    return fm.encode_image(img)

def halves(n=2):
    cur = torch.cuda.current_stream()
    outs = []
    streams = [s1, s2][:n]
    parts = img.chunk(n)
    for st, part in zip(streams, parts):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(fm.encode_image(part))
    for st in streams:
        cur.wait_stream(st)
    return torch.cat(outs)

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

a = whole(); b = halves()
print("max |diff|", (a - b).abs().max().item())
print(f"one batch {timeit(whole):.2f} ms; two halves on two streams {timeit(halves):.2f} ms  (SL_G8_MIN_PCT={os.environ.get('SL_G8_MIN_PCT','50')})")
