"""Target for rocprofv3 --pmc passes over the cosine GEMM only (configs[3] shapes), both arithmetic modes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402


def main():
    q = torch.randn(10000, 1152, device="cuda:0")
    y = torch.randn(9216, 1152, device="cuda:0")
    for mode in ("bf16x3", "f32"):
        N.set_gemm_mode(mode)
        for _ in range(4):
            N.similarity(q, y)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
