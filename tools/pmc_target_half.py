"""Target for the rocprofv3 --pmc passes over K1 on 2-byte activations (`rowreduce_h`: the bench's `half_precision_model` leg):
ResNet-50 layer2/3/4 outputs at B = 256 in fp16, rotating over enough copies that the 256 MiB Infinity Cache cannot serve the reads.
    rocprofv3 --pmc FETCH_SIZE -- python tools/pmc_target_half.py      (and a second pass with WRITE_SIZE)
    python tools/pmc_traffic_half.py <fetch.csv> <write.csv>"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
REPS = 6


def main():
    for (B, C, H, W) in ((256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7)):
        xs = [torch.randn(B, C, H, W, device=DEV).relu_().half() for _ in range(6)]
        cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
        for i in range(REPS):
            N.reduce_conv(xs[i % 6], N.SL_CONV_MAX, cand, None)
        torch.cuda.synchronize()
        del xs


if __name__ == "__main__":
    main()
