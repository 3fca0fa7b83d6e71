import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from semanticlens_amd import _native as N
dev = torch.device("cuda:0")
for i in range(4):
    r = bench.probing_leg(dev)
    print(i, round(r["roofline"]["achieved"],1), round(r["roofline"]["frac"],3), round(r["fp32_mfma_mode"]["roofline"]["frac"],3), round(r["value"]), flush=True)
r = bench.probing_leg(dev)
print("f32 mode Msim/s", round(r["fp32_mfma_mode"]["value"]), "wall ms", round(r["fp32_mfma_mode"]["wall_ms"], 3), "| bf16x3 wall ms", round(r["wall_ms"], 3))
