"""Round 6 A/B of `attention_bf16x3_kernel`'s launch shapes on long sequences (lab option "attn_split": 1 = one 8-wave workgroup per
(image, head) with 128-key chunks — rounds 3-5 —, 2 = one 4-wave workgroup per round of four query tiles with 64-key chunks, two
workgroups per CU): time per call, equality of the two results (same arithmetic per element, another chunking of the online softmax:
equal to ~1e-6, not bitwise), distance from float64, and the so400m tower around it."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def ref64(qkv, B, T, H, D):
    q, k, v = qkv.double().reshape(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) / D ** 0.5, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * T, H * D)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for B, T, H, D in ((256, 256, 16, 72), (64, 256, 16, 72), (64, 197, 12, 64), (32, 577, 16, 64), (8, 729, 16, 72), (16, 257, 16, 80)):
    g = torch.Generator(device=DEV).manual_seed(T)
    qkv = torch.randn(B * T, 3 * H * D, device=DEV, generator=g)
    outs, us = {}, {}
    for mode in (1, 2):
        N.set_option("attn_split", mode)
        out = torch.empty(B * T, H * D, device=DEV)
        us[mode] = timed(lambda: N.attention(qkv, B, T, H, D, False, out=out, bf16x3=True))
        outs[mode] = out.clone()
    N.set_option("attn_split", 0)
    small = min(B, 4)
    want = ref64(qkv[: small * T], small, T, H, D)
    err = {m: float((outs[m][: small * T].double() - want).abs().max()) for m in (1, 2)}
    print(f"B={B} T={T} H={H} D={D}: one workgroup per head {us[1]:8.1f} us | split {us[2]:8.1f} us ({us[1] / us[2]:.2f}x) | "
          f"max |a - b| {float((outs[1] - outs[2]).abs().max()):.2e} | vs float64 {err[1]:.2e} / {err[2]:.2e}", flush=True)

# the tower around it
import synth  # noqa: E402
from semanticlens_amd.foundation_models import NativeSigLip  # noqa: E402

fm = NativeSigLip(synth.SyntheticSigLip(device=DEV))
for Bq in (64, 256):
    x = torch.randn(Bq, 3, 224, 224, device=DEV)
    line = []
    for mode in (1, 2, 1, 2):
        N.set_option("attn_split", mode)
        for _ in range(2):
            fm.encode_image(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            fm.encode_image(x)
        torch.cuda.synchronize()
        line.append(f"mode {mode}: {(time.perf_counter() - t) / 4 * 1e3:.1f} ms")
    N.set_option("attn_split", 0)
    print(f"so400m image tower B={Bq}: " + " | ".join(line), flush=True)
