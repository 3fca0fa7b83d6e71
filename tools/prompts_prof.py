import cProfile, pstats, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth
from semanticlens_amd import Lens
from semanticlens_amd.foundation_models.native_clip import NativeClip
dev = torch.device("cuda:0")
fm = NativeClip(synth.SyntheticClip(device=dev))
words = ["zebra", "stripe", "wheel", "sky", "grass", "dog", "cat", "red", "round", "metal", "wood", "water", "face", "text"]
prompts = [f"a photo of a {words[i % 14]} {words[(i // 14) % 14]} {i}" for i in range(10000)]
g = torch.Generator(device=dev).manual_seed(3)
db = {f"block{i}": torch.randn(768, 512, device=dev, generator=g) for i in range(12)}
lens = Lens(fm, device=dev)
lens.text_probing(prompts[:1024], db, batch_size=1024); torch.cuda.synchronize()
for bs in (1024, 2500, 10000):
    lens.text_probing(prompts, db, batch_size=bs); torch.cuda.synchronize()
    t0 = time.perf_counter(); lens.text_probing(prompts, db, batch_size=bs); torch.cuda.synchronize()
    print("batch", bs, f"{1e3*(time.perf_counter()-t0):.1f} ms")
t0 = time.perf_counter(); toks = [fm.tokenize(prompts[s:s+1024]) for s in range(0, 10000, 1024)]; torch.cuda.synchronize()
print("tokenize only", f"{1e3*(time.perf_counter()-t0):.1f} ms")
t0 = time.perf_counter()
for t in toks: fm.encode_text(t)
torch.cuda.synchronize(); print("towers only", f"{1e3*(time.perf_counter()-t0):.1f} ms")
pr = cProfile.Profile(); pr.enable(); lens.text_probing(prompts, db, batch_size=1024); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
