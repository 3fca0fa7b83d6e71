"""Kernel-only micro-benchmarks (SURVEY.md §8d micro-inputs): HBM GB/s of the reduce kernels at the
BASELINE config shapes, merge-kernel latency, TFLOP/s of the cosine GEMM.  GPU only."""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402
from semanticlens_amd.component_visualization.activation_caching import ActMax  # noqa: E402

DEV = "cuda:0"


def timed(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench_reduce(B, C, H, W, agg=N.SL_CONV_MAX, channels_last=False, flush_mb=512):
    x = torch.randn(B, C, H, W, device=DEV).relu_()
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    cand = torch.empty((B, C), dtype=torch.bfloat16, device=DEV)
    # rotate over several copies so the input does not sit in the 256 MiB Infinity Cache
    nbytes = x.numel() * 4
    ncopy = max(1, min(8, int(flush_mb * 2**20 // nbytes) + 1))
    xs = [x] + [x.clone() for _ in range(ncopy - 1)]
    i = [0]

    def run():
        N.reduce_conv(xs[i[0] % ncopy], agg, cand, None)
        i[0] += 1

    ms = timed(run, iters=4 * ncopy)
    return {"shape": [B, C, H, W], "cl": channels_last, "agg": agg, "ms": ms, "GBps": nbytes / ms / 1e6, "copies": ncopy}


def bench_tokens(B, T, F, agg=N.SL_TOK_MAX):
    x = torch.randn(B, T, F, device=DEV)
    cand = torch.empty((B, F), dtype=torch.bfloat16, device=DEV)
    nbytes = x.numel() * 4
    ncopy = max(1, min(8, int(512 * 2**20 // nbytes) + 1))
    xs = [x] + [x.clone() for _ in range(ncopy - 1)]
    i = [0]

    def run():
        N.reduce_tokens(xs[i[0] % ncopy], agg, 0, cand, None)
        i[0] += 1

    ms = timed(run, iters=4 * ncopy)
    return {"shape": [B, T, F], "agg": agg, "ms": ms, "GBps": nbytes / ms / 1e6}


def bench_merge(C, k, B, mode, steady=True):
    am = ActMax(k, C, tie_mode=mode)
    g = torch.Generator(device=DEV).manual_seed(0)
    # warm the state with 50 batches so the steady-state filter rate applies
    for s in range(50 if steady else 0):
        am.update(torch.randn(B, C, device=DEV, generator=g).relu_(), torch.arange(s * B, (s + 1) * B))
    acts = torch.randn(B, C, device=DEV, generator=g).relu_()
    ids = torch.arange(10**6, 10**6 + B, device=DEV)
    ms = timed(lambda: am.update(acts, ids), iters=20)
    return {"C": C, "k": k, "B": B, "mode": mode, "steady": steady, "ms_update_incl_cast": ms}


def bench_gemm(Q, C, D):
    x = torch.randn(Q, D, device=DEV)
    y = torch.randn(C, D, device=DEV)
    ms = timed(lambda: N.similarity(x, y), iters=10)
    return {"Q": Q, "C": C, "D": D, "ms": ms, "TFLOPs": 2.0 * Q * C * D / ms / 1e9, "Msim_per_s": Q * C / ms / 1e3}


def bench_preprocess(B, h, w, S=224, cpu_images=16):
    """K12 on a batch of (h, w) RGB images (resident on the device) vs the oracle (Pillow's algorithm) on one core."""
    import numpy as np

    px = torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8, device=DEV)
    plan, info = N.preprocess_plan([(h, w)] * B, S)
    plan_d = plan.to(DEV)
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    flat = px.reshape(-1)
    ms = timed(lambda: N.preprocess(flat, plan_d, info, S, mean, std), iters=10)
    alg = B * (h * w * 3 + 3 * S * S * 4)
    res = {"B": B, "hw": [h, w], "S": S, "ms": ms, "images_per_s": B / ms * 1e3, "GBps_algorithmic": alg / ms / 1e6}
    try:
        import oracle

        imgs = px[:cpu_images].cpu().numpy()
        t0 = time.perf_counter()
        for i in range(cpu_images):
            oracle.preprocess(imgs[i], S, mean, std)
        res["cpu_oracle_images_per_s_1core"] = cpu_images / (time.perf_counter() - t0)
    except Exception as e:  # the oracle is optional here
        res["cpu_oracle"] = repr(e)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default=None, help="run one family: preprocess | gemm | reduce")
    args = ap.parse_args()
    out = []
    if args.only == "gemm":
        for q, c, d in ((10000, 768, 1152), (10000, 9216, 1152), (4096, 2048, 512), (12800, 768, 768), (12800, 3072, 768), (12800, 768, 3072)):
            print(json.dumps(("gemm", bench_gemm(q, c, d))), flush=True)
        return
    if args.only == "preprocess":
        for B, h, w in ((256, 500, 375), (256, 375, 500), (256, 224, 224), (64, 1200, 1600)):
            print(json.dumps(("preprocess", bench_preprocess(B, h, w))), flush=True)
        return
    if args.only == "reduce":
        for s in ((256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7), (64, 2048, 7, 7), (256, 192, 56, 56), (256, 384, 28, 28), (256, 1536, 7, 7)):
            print(json.dumps(("reduce_max", bench_reduce(*s))), flush=True)
        print(json.dumps(("reduce_mean", bench_reduce(256, 2048, 7, 7, agg=N.SL_CONV_MEAN))), flush=True)
        print(json.dumps(("tokens_max", bench_tokens(256, 197, 768))), flush=True)
        return
    shapes = [(256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7), (64, 512, 28, 28), (64, 1024, 14, 14), (64, 2048, 7, 7)]
    if not args.quick:
        shapes += [(256, 192, 56, 56), (256, 1536, 7, 7), (32, 2048, 7, 7), (1024, 2048, 7, 7)]
    for s in shapes:
        out.append(("reduce_max", bench_reduce(*s)))
        print(json.dumps(out[-1]), flush=True)
    out.append(("reduce_mean", bench_reduce(256, 2048, 7, 7, agg=N.SL_CONV_MEAN)))
    print(json.dumps(out[-1]), flush=True)
    out.append(("reduce_max_cl", bench_reduce(256, 2048, 7, 7, channels_last=True)))
    print(json.dumps(out[-1]), flush=True)
    out.append(("reduce_max_cl", bench_reduce(256, 512, 28, 28, channels_last=True)))
    print(json.dumps(out[-1]), flush=True)
    out.append(("tokens_max", bench_tokens(256, 197, 768)))
    print(json.dumps(out[-1]), flush=True)
    for mode in ("total", "aten"):
        for C, k, B in ((2048, 20, 256), (512, 20, 256), (2048, 100, 256), (2048, 20, 64)):
            out.append(("merge", bench_merge(C, k, B, mode)))
            print(json.dumps(out[-1]), flush=True)
    out.append(("merge_cold", bench_merge(2048, 20, 256, "total", steady=False)))
    print(json.dumps(out[-1]), flush=True)
    for q, c, d in ((10000, 768, 1152), (10000, 9216, 1152), (4096, 2048, 512)):
        out.append(("gemm", bench_gemm(q, c, d)))
        print(json.dumps(out[-1]), flush=True)
    for B, h, w in ((256, 500, 375), (256, 224, 224)):
        print(json.dumps(("preprocess", bench_preprocess(B, h, w))), flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"done in {time.time() - t0:.1f}s")
