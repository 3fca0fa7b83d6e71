"""Target for the rocprofv3 --pmc passes over K2 / the component-contiguous K1 kernel (colreduce2): the configs[3] block shape
in fp32 and fp16 and the four ConvNeXt-L stage outputs (channels_last), cold (rotating over > 1.2 GB), 6 launches each."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
REPS = 6
SHAPES = [((256, 197, 768), torch.float32), ((256, 197, 768), torch.float16), ((256, 3136, 192), torch.float32),
          ((256, 784, 384), torch.float32), ((256, 196, 768), torch.float32), ((256, 49, 1536), torch.float32)]


def main():
    N.set_reduce_policy(0, 0)
    for shape, dtype in SHAPES:
        nbytes = shape[0] * shape[1] * shape[2] * torch.finfo(dtype).bits // 8
        nbuf = max(2, int(1.2e9 // nbytes))
        xs = [torch.randn(shape, device=DEV, dtype=torch.float32).to(dtype) for _ in range(nbuf)]
        cand = torch.empty((shape[0], shape[2]), dtype=torch.bfloat16, device=DEV)
        for i in range(REPS):
            N.reduce_tokens(xs[i % nbuf], N.SL_TOK_MAX, 0, cand, None)
        torch.cuda.synchronize()
        del xs
    # round 5: the table form — twelve (256, 197, 768) block outputs in ONE launch (sl_reduce_tokens_multi), 1.86 GB
    xs = [torch.randn((256, 197, 768), device=DEV) for _ in range(12)]
    cand = torch.empty((12, 256, 768), dtype=torch.bfloat16, device=DEV)
    for _ in range(REPS):
        N.reduce_multi("tokens", xs, N.SL_TOK_MAX, 0, cand)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
