"""K2 (token / channels_last reduce) variants, cold and behind a producer, one process per variant (the library reads
SL_COLREDUCE_* once): the VGPR-load kernel of rounds 1-3 against the LDS-DMA ring kernel of round 4 at ring depths 2-4 and
4 / 8 waves per task.  Correctness of every variant against torch first.

    python tools/k2_lab.py            # the sweep (spawns itself per variant)
    python tools/k2_lab.py one        # one variant: environment as given
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SHAPES = [((256, 197, 768), "float32"), ((256, 197, 768), "bfloat16"), ((256, 197, 768), "float16"), ((256, 50, 768), "float32"), ((48, 729, 1152), "float32"),
          ((48, 729, 1152), "float16"), ((64, 256, 1152), "float32"), ((256, 196, 1024), "float32"), ((256, 257, 1024), "bfloat16"),
          # ConvNeXt-L stage outputs as the model emits them (channels_last: the residual add takes the permuted branch's layout)
          ((256, 3136, 192), "float32"), ((256, 784, 384), "float32"), ((256, 196, 768), "float32"), ((256, 49, 1536), "float32")]


def one():
    import torch

    from semanticlens_amd import _native as N

    dev = "cuda:0"
    out = []
    for shape, dname in SHAPES:
        dtype = getattr(torch, dname)
        nbytes = shape[0] * shape[1] * shape[2] * torch.finfo(dtype).bits // 8
        nbuf = max(2, int(1.3e9 // nbytes))
        xs = [torch.randn(shape, device=dev, dtype=torch.float32).to(dtype) for _ in range(nbuf)]
        cand = torch.empty((shape[0], shape[2]), dtype=torch.bfloat16, device=dev)
        f32 = torch.empty((shape[0], shape[2]), dtype=torch.float32, device=dev)
        # correctness: max / absmax exact, mean within fp32 summation order
        x = xs[0].clone()
        x[3, 5, 7] = float("nan")
        x[4, :, 9] = float("-inf")
        for agg, ref in ((N.SL_TOK_MAX, lambda t: t.amax(1)), (N.SL_TOK_ABSMAX, lambda t: t.abs().amax(1)), (N.SL_TOK_MEAN, lambda t: t.mean(1))):
            N.reduce_tokens(x, agg, 0, cand, f32)
            want = ref(x).float()
            if agg == N.SL_TOK_MEAN:
                ok = torch.allclose(f32, want, rtol=2e-3 if dtype != torch.float32 else 1e-5, atol=1e-5, equal_nan=True)
            else:
                ok = torch.equal(torch.nan_to_num(f32, nan=1e30), torch.nan_to_num(want, nan=1e30))
            assert ok, (shape, dname, agg)
        res = {}
        for regime in ("cold", "pipe"):
            N.set_reduce_policy(0, 0) if regime == "cold" else N.set_reduce_policy(None, None)
            for x_ in xs[:2]:
                N.reduce_tokens(x_, N.SL_TOK_MAX, 0, cand, None)
            torch.cuda.synchronize()
            N.prof_enable(True)
            N.prof_reset()
            for i in range(3 * nbuf if regime == "cold" else 24):
                x_ = xs[i % nbuf]
                if regime == "pipe":  # a residual add writes the input right before the reduce (what a transformer block ends with)
                    torch.add(x_, 0.5, out=x_)
                N.reduce_tokens(x_, N.SL_TOK_MAX, 0, cand, None)
            ms, n, nb = N.prof_read(N.SL_PROF_REDUCE)
            N.prof_enable(False)
            res[regime] = (nb / ms / 1e6, ms / n * 1e3)
        out.append(f"{str(shape):16s} {dname:8s} {nbytes / 1e6:6.0f} MB  cold {res['cold'][0]:6.0f} GB/s {res['cold'][1]:6.1f} us   "
                   f"pipe {res['pipe'][0]:6.0f} GB/s {res['pipe'][1]:6.1f} us")
    print("\n".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        variants = [dict(SL_COLREDUCE_IMPL="vgpr"), dict(SL_COLREDUCE_IMPL="v2"), dict(SL_COLREDUCE_IMPL="v2", SL_COLREDUCE_NW="4"),
                    dict(SL_COLREDUCE_IMPL="v2", SL_COLREDUCE_NW="8"), dict(SL_COLREDUCE_IMPL="v2", SL_COLREDUCE_LPR="64"),
                    dict(SL_COLREDUCE_IMPL="v2", SL_COLREDUCE_LPR="32"), dict(SL_COLREDUCE_IMPL="dma", SL_COLREDUCE_DEPTH="2", SL_COLREDUCE_NW="8"),
                    dict(SL_COLREDUCE_IMPL="dma", SL_COLREDUCE_DEPTH="3", SL_COLREDUCE_NW="4")]
        if len(sys.argv) > 1:
            variants = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
        for v in variants:
            env = {k: val for k, val in os.environ.items() if not k.startswith("SL_COLREDUCE_")}
            env.update(v)
            print("=== " + " ".join(f"{k[13:].lower()}={val}" for k, val in v.items()), flush=True)
            subprocess.run([sys.executable, __file__, "one"], env=env, check=False)
