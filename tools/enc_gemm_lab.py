"""Per-kernel times of the native ViT-B/32 image tower at B = 256 (M = 12 800 tokens), kernel by kernel and epilogue by
epilogue: which of the four GEMMs of a block costs what, and what their epilogues (bias / residual / GELU + split output)
add on top of the bare product.  Per-dispatch times from sl_prof (HIP events stamped by the dispatch itself)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
M, W, F = int(sys.argv[1]) if len(sys.argv) > 1 else 12800, 768, 3072
g = torch.Generator(device=DEV).manual_seed(0)


def rnd(*s):
    return torch.randn(*s, device=DEV, generator=g)


def timed(fn, fam=N.SL_PROF_GEMM, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    N.prof_enable(True)
    N.prof_reset()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, n, work = N.prof_read(fam)
    N.prof_enable(False)
    return ms / max(n, 1) * 1e3, n // reps


x, hid = rnd(M, W), rnd(M, F)
sx, shid = N.Split.of(x), N.Split.of(hid)
w_qkv, w_o, w_fc, w_pr = rnd(3 * W, W) * 0.03, rnd(W, W) * 0.03, rnd(F, W) * 0.03, rnd(W, F) * 0.02
s_qkv, s_o, s_fc, s_pr = (N.Split.of(w) for w in (w_qkv, w_o, w_fc, w_pr))
b_qkv, b_o, b_fc, b_pr = rnd(3 * W), rnd(W), rnd(F), rnd(W)
qkv, res = torch.empty(M, 3 * W, device=DEV), rnd(M, W)
o_split, hid_split = N.Split(M, W, DEV), N.Split(M, F, DEV)
out_w = torch.empty(M, W, device=DEV)
out_f = torch.empty(M, F, device=DEV)

cases = [
    ("qkv   (M x 2304 x 768) bias -> f32", lambda: N.linear3(sx, s_qkv, b_qkv, out=qkv), 2 * M * 3 * W * W),
    ("qkv   bare                       ", lambda: N.linear3(sx, s_qkv, out=qkv), 2 * M * 3 * W * W),
    ("oproj (M x 768 x 768) bias + residual in place", lambda: N.linear3(sx, s_o, b_o, residual=res, out=res), 2 * M * W * W),
    ("oproj bare -> f32                ", lambda: N.linear3(sx, s_o, out=out_w), 2 * M * W * W),
    ("fc1   (M x 3072 x 768) bias + GELU -> split", lambda: N.linear3(sx, s_fc, b_fc, act=N.SL_ACT_GELU, out_split=hid_split), 2 * M * F * W),
    ("fc1   bias -> split (no act)     ", lambda: N.linear3(sx, s_fc, b_fc, out_split=hid_split), 2 * M * F * W),
    ("fc1   bare -> f32                ", lambda: N.linear3(sx, s_fc, out=out_f), 2 * M * F * W),
    ("fc2   (M x 768 x 3072) bias + residual in place", lambda: N.linear3(shid, s_pr, b_pr, residual=res, out=res), 2 * M * W * F),
    ("fc2   bare -> f32                ", lambda: N.linear3(shid, s_pr, out=out_w), 2 * M * W * F),
]
tot = 0.0
for name, fn, flops in cases:
    us, n = timed(fn)
    print(f"{name:52s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  ({n} launch)")
    if "bare" not in name and "no act" not in name:
        tot += us
print(f"four GEMMs of a block: {tot:.1f} us  -> x12 = {tot * 12 / 1e3:.2f} ms")

import time  # noqa: E402

gam, bet = rnd(W), rnd(W)
h = N.Split(M, W, DEV)


def wall(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


print(f"layernorm -> split: {wall(lambda: N.layernorm(x, gam, bet, 1e-5, out_split=h)):.1f} us (wall, back to back)")
T = 50
B = M // T
print(f"attention (B={B}, T={T}, 12 x 64) -> split: {wall(lambda: N.attention(qkv, B, T, 12, 64, False, out_split=o_split)):.1f} us (fp32 MFMA)  "
      f"{wall(lambda: N.attention(qkv, B, T, 12, 64, False, out_split=o_split, bf16x3=True)):.1f} us (bf16x3)")
