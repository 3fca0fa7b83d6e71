"""Does giving the embed stream its own CUs help the concept-DB step?  The step of bench.py (ResNet-50 forward + collect on
one HIP stream, native ViT-B/32 embed on a second) with the two streams created by `hipExtStreamCreateWithCUMask` over
disjoint CU sets: bits [0, 256 - n) for forward + collect, the last n bits for the embed.  n = 0: unmasked streams."""
import ctypes
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import synth  # noqa: E402
from semanticlens_amd import _native as N  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
DEV = torch.device("cuda:0")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (w * 32 + b) in bits) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=DEV)


def main():
    B, nb, k = 256, 24, 20
    torch.manual_seed(0)
    model = synth.resnet50().to(DEV).eval()
    fm = NativeClip(synth.SyntheticClip(device=DEV), gemm="bf16x3")
    ids = [torch.arange(s * B, (s + 1) * B, device=DEV) for s in range(6)]
    pool = [synth.synth_images_u8(i) for i in ids]
    batches = [pool[s % 6] for s in range(nb)]
    splits = [int(a) for a in sys.argv[1:]] or [0, 32, 48, 64, 96]
    for how in ("interleaved", "contiguous"):
        for n in splits:
            if n == 0:
                if how == "contiguous":
                    continue
                main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream()
            elif how == "contiguous":
                main_s, side_s = masked_stream(set(range(0, 256 - n))), masked_stream(set(range(256 - n, 256)))
            else:  # every (256 / n)-th bit
                step = 256 // n
                enc = set(range(step - 1, 256, step))
                main_s, side_s = masked_stream(set(range(256)) - enc), masked_stream(enc)
            best = None
            for rep in range(4):
                cv = bench.make_cv(model, nb * B, k, "aten")
                N.prof_enable(True)
                N.prof_reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.cuda.stream(main_s):
                    bench_side = torch.cuda.Stream
                    torch.cuda.Stream = lambda *a, **kw: side_s  # run_steps creates its side stream: hand it ours
                    try:
                        emb = bench.run_steps(cv, fm, batches, 0, nb * B)
                    finally:
                        torch.cuda.Stream = bench_side
                    bench.finish_job(cv, emb, 0, nb * B, False)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                red_ms, red_n, red_bytes = N.prof_read(N.SL_PROF_REDUCE)
                N.prof_enable(False)
                if rep and (best is None or dt < best[0]):
                    best = (dt, red_bytes / red_ms / 1e9 / 8.0)
            print(f"{how:12s} embed CUs {n:3d}: {nb * B / best[0]:7.0f} images/s  {best[0] / nb * 1e3:6.2f} ms/batch   K1 {best[1]:.3f} of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
