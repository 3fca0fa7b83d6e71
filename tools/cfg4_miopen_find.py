import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth
torch.backends.cudnn.benchmark = sys.argv[1] == "1"
m = synth.convnext_l().to("cuda:0")
x = torch.randn(256, 3, 224, 224, device="cuda:0")
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): m(x)
    torch.cuda.synchronize()
print("benchmark", sys.argv[1], "forward ms", (time.perf_counter() - t) / 5 * 1e3)
