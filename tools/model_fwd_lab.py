"""What the probed model's forward (80 % of a bench step) costs under PyTorch-level settings: MIOpen find mode,
channels_last.  python tools/model_fwd_lab.py  (GPU)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth  # noqa: E402


def run(tag, find, cl, B=256, iters=8):
    torch.backends.cudnn.benchmark = find
    m = synth.resnet50().cuda().eval()
    x = torch.randn(B, 3, 224, 224, device="cuda")
    if cl:
        m = m.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        t0 = time.time()
        for _ in range(2):
            y = m(x)
        torch.cuda.synchronize()
        warm = time.time() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            y = m(x)
        e1.record()
        torch.cuda.synchronize()
    print(f"{tag:36s} {e0.elapsed_time(e1) / iters:7.2f} ms / {B} images   (warm-up {warm:.1f} s)  checksum {float(y.float().abs().mean()):.6f}", flush=True)


if __name__ == "__main__":
    run("default", False, False)
    run("channels_last", False, True)
    run("miopen find", True, False)
    run("miopen find + channels_last", True, True)
