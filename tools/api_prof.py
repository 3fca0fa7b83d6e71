import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
import bench, synth
from types import SimpleNamespace
dev = torch.device("cuda:0")
model = synth.resnet50().to(dev)
fm_base = synth.SyntheticClip(device=dev)
args = SimpleNamespace(api_images=1024, batch=256, k=20)
bench.api_path_leg(dev, model, fm_base, args)  # warm
pr = cProfile.Profile(); pr.enable()
r = bench.api_path_leg(dev, model, fm_base, args)
pr.disable()
print(r["api_path_images_per_s"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
