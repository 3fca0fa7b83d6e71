"""Round 6: the QKV projection + attention of one tower block on long non-causal sequences, two routes:
  fp32 route   sl_linear_bf16x3 (qkv fp32) + sl_attention_bf16x3 (K / V converted and transposed through registers per workgroup)
  image route  sl_linear_bf16x3_qkv (K / V written as the attention kernel's LDS image) + sl_attention_bf16x3_kv (LDS-DMA staging)
Bitwise equality of the attention output, time of each call, and the so400m tower with either route."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=10):
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


for B, T, H, D in ((256, 256, 16, 72), (64, 256, 16, 72), (64, 197, 12, 64), (8, 729, 16, 72), (16, 257, 16, 80), (3, 130, 2, 32), (5, 161, 3, 96)):
    W = H * D
    g = torch.Generator(device=DEV).manual_seed(T + D)
    x = torch.randn(B * T, W, device=DEV, generator=g)
    w = torch.randn(3 * W, W, device=DEV, generator=g) * W ** -0.5
    bias = torch.randn(3 * W, device=DEV, generator=g) * 0.1
    sx, sw = N.Split.of(x), N.Split.of(w)
    qkv = torch.empty(B * T, 3 * W, device=DEV)
    out_a = torch.empty(B * T, W, device=DEV)
    q = torch.empty(B * T, W, device=DEV)
    kv = N.kv_image(B, T, H, D, DEV)
    out_b = torch.empty(B * T, W, device=DEV)
    t_lin_a = timed(lambda: N.linear3(sx, sw, bias, out=qkv))
    t_att_a = timed(lambda: N.attention(qkv, B, T, H, D, False, out=out_a, bf16x3=True))
    t_lin_b = timed(lambda: N.linear3_qkv(sx, sw, bias, B, T, H, D, q, kv))
    t_att_b = timed(lambda: N.attention_kv(q, kv, B, T, H, D, out=out_b))
    same_q = torch.equal(q, qkv[:, :W])
    same = torch.equal(out_a, out_b)
    print(f"B={B} T={T} H={H} D={D}: projection {t_lin_a:8.1f} -> {t_lin_b:8.1f} us | attention {t_att_a:8.1f} -> {t_att_b:8.1f} us | "
          f"sum {t_lin_a + t_att_a:8.1f} -> {t_lin_b + t_att_b:8.1f} | Q equal {same_q} | attention output bit-identical {same}"
          + ("" if same else f" (max diff {float((out_a - out_b).abs().max()):.3e})"), flush=True)

import synth  # noqa: E402
from semanticlens_amd.foundation_models import NativeSigLip, native_clip  # noqa: E402

fm = NativeSigLip(synth.SyntheticSigLip(device=DEV))
for Bq in (64, 256):
    x = torch.randn(Bq, 3, 224, 224, device=DEV)
    line, feats = [], {}
    for route in (False, True, False, True):
        native_clip.KV_ROUTE = route
        for _ in range(2):
            f = fm.encode_image(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            f = fm.encode_image(x)
        torch.cuda.synchronize()
        feats[route] = f.clone()
        line.append(f"{'image' if route else 'fp32'} route {(time.perf_counter() - t) / 4 * 1e3:.1f} ms")
    native_clip.KV_ROUTE = True
    print(f"so400m image tower B={Bq}: " + " | ".join(line) + f" | features bit-identical {torch.equal(feats[False], feats[True])}", flush=True)
