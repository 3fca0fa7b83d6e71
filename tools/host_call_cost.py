"""Host cost of the pieces of one native call (us): device guard, stream lookup, pointer boxing, a ctypes call, torch.empty."""
import ctypes
import sys
import timeit
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

dev = torch.device("cuda:0")
x = torch.empty(64, 64, device=dev)
g = {"torch": torch, "N": N, "dev": dev, "x": x, "ctypes": ctypes}


def us(stmt, n=20000):
    return timeit.timeit(stmt, globals=g, number=n) / n * 1e6


GUARD = """
with torch.cuda.device(dev):
    pass"""
print(f"with torch.cuda.device(dev): pass      {us(GUARD):.2f}")
print(f"torch.cuda.current_device()             {us('torch.cuda.current_device()'):.2f}")
print(f"torch.cuda.current_stream(dev).cuda_stream {us('torch.cuda.current_stream(dev).cuda_stream'):.2f}")
print(f"N._stream(x)                            {us('N._stream(x)'):.2f}")
print(f"N._ptr(x)                               {us('N._ptr(x)'):.2f}")
print(f"torch.empty((64, 64), device=dev)       {us('torch.empty((64, 64), dtype=torch.float32, device=dev)'):.2f}")
print(f"N.lib().sl_last_error()                 {us('N.lib().sl_last_error()'):.2f}")
sx, sw = N.Split.of(torch.randn(50, 768, device=dev)), N.Split.of(torch.randn(768, 768, device=dev))
out = torch.empty(50, 768, device=dev)
g.update(sx=sx, sw=sw, out=out)
print(f"N.linear3(sx, sw, out=out) (enqueue)    {us('N.linear3(sx, sw, out=out)', 5000):.2f}")
torch.cuda.synchronize()
