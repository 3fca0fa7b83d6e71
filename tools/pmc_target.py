"""Target for the rocprofv3 --pmc passes: launches only the hot-path kernels at the bench shapes
(ResNet-50 layer2/3/4, B=256, fp32; cosine GEMM at configs[3] shapes) a few times each, rotating over
enough input copies that the 256 MiB Infinity Cache cannot serve the reads."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
REPS = 6


def main():
    for (B, C, H, W) in ((256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7)):
        xs = [torch.randn(B, C, H, W, device=DEV).relu_() for _ in range(4)]
        cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
        for i in range(REPS):
            N.reduce_conv(xs[i % 4], N.SL_CONV_MAX, cand, None)
        torch.cuda.synchronize()
        del xs
    q = torch.randn(10000, 1152, device=DEV)
    y = torch.randn(768, 1152, device=DEV)
    for _ in range(REPS):
        N.similarity(q, y)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
