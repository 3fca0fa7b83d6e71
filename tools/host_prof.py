import cProfile, pstats, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth
from semanticlens_amd.foundation_models.native_clip import NativeClip
dev = torch.device("cuda:0")
fm = NativeClip(synth.SyntheticClip(device=dev))
u8 = synth.synth_images_u8(torch.arange(256, device=dev))
with torch.no_grad():
    for _ in range(3):
        fm.encode_image(fm.preprocess(u8))
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        fm.encode_image(fm.preprocess(u8))
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
