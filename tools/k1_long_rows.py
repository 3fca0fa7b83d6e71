"""K1 on rows longer than an LDS-DMA batch (> 4 KiB: 56 x 56 maps and larger): correctness against torch, then cold (inputs rotated
through > 1.3 GB, read-once policy) and behind a producer (in-place ReLU).  Round 4 ran it with SL_ROWREDUCE_LONG=0 / 1 to compare
rowreduce_fast (fp32) / rowreduce_h (half precision) with a ping-pong stream kernel built for such rows; the new kernel lost
(profiles/r04_k1_long_rows.txt) and was removed from the library, so both settings now run the same kernels.

    python tools/k1_long_rows.py
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SHAPES = [((256, 192, 56, 56), "float32"), ((256, 192, 56, 56), "float16"), ((256, 192, 56, 56), "bfloat16"), ((256, 64, 112, 112), "float32"),
          ((64, 256, 56, 56), "float32"), ((128, 96, 40, 40), "float32"), ((32, 40, 113, 108), "float32"), ((256, 64, 112, 112), "float16")]


def one():
    import torch

    from semanticlens_amd import _native as N

    dev = "cuda:0"
    for shape, dname in SHAPES:
        dtype = getattr(torch, dname)
        nbytes = torch.tensor(shape).prod().item() * torch.finfo(dtype).bits // 8
        nbuf = max(2, int(1.3e9 // nbytes))
        xs = [torch.randn(shape, device=dev, dtype=torch.float32).to(dtype) for _ in range(nbuf)]
        cand = torch.empty(shape[:2], dtype=torch.bfloat16, device=dev)
        f32 = torch.empty(shape[:2], dtype=torch.float32, device=dev)
        x = xs[0].clone()
        x[1, 3, 5, 7] = float("nan")
        x[2, 4] = float("-inf")
        x[2, 5, 0, 0], x[2, 5, 1, 1] = float("inf"), float("-inf")
        for agg, ref in ((N.SL_CONV_MAX, lambda t: t.flatten(2).amax(-1)), (N.SL_CONV_MEAN, lambda t: t.flatten(2).mean(-1)),
                         (N.SL_CONV_SUM, lambda t: t.float().flatten(2).sum(-1))):
            N.reduce_conv(x, agg, cand, f32)
            want = ref(x).float()
            if agg == N.SL_CONV_MAX:
                ok = torch.equal(torch.nan_to_num(f32, nan=1e30), torch.nan_to_num(want, nan=1e30))
                ok = ok and torch.equal(cand.view(torch.int16), want.to(torch.bfloat16).view(torch.int16))
            else:
                fin = torch.isfinite(want)
                ok = torch.allclose(f32[fin], want[fin], rtol=1e-2 if dtype != torch.float32 else 2e-5, atol=1e-4) and \
                    torch.equal(torch.isnan(f32), torch.isnan(want))
            assert ok, (shape, dname, agg)
        res = {}
        for regime in ("cold", "pipe"):
            N.set_reduce_policy(0, 0) if regime == "cold" else N.set_reduce_policy(None, None)
            for x_ in xs[:2]:
                N.reduce_conv(x_, N.SL_CONV_MAX, cand, None)
            torch.cuda.synchronize()
            N.prof_enable(True)
            N.prof_reset()
            for i in range(3 * nbuf if regime == "cold" else 16):
                x_ = xs[i % nbuf]
                if regime == "pipe":
                    torch.relu_(x_)
                N.reduce_conv(x_, N.SL_CONV_MAX, cand, None)
            ms, n, nb = N.prof_read(N.SL_PROF_REDUCE)
            N.prof_enable(False)
            res[regime] = (nb / ms / 1e6, ms / n * 1e3)
        print(f"{str(shape):20s} {dname:8s} {nbytes / 1e6:6.0f} MB  cold {res['cold'][0]:6.0f} GB/s {res['cold'][1]:6.1f} us   "
              f"pipe {res['pipe'][0]:6.0f} GB/s {res['pipe'][1]:6.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for flag in ("0", "1"):
            print(f"=== SL_ROWREDUCE_LONG={flag}", flush=True)
            subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, SL_ROWREDUCE_LONG=flag), check=False)
