"""Golden vectors for the image preprocessing stage (foundation_models/clip.py:157-163 -> open_clip transform).

Run in the build container (needs Pillow; 12.2.0 here):  python tests/golden/make_golden_preprocess.py
The resize + centre-crop bytes come from Pillow itself (`Image.resize`, `Image.crop` with torchvision's published
size/offset rules); the normalised floats from torch's own `div` / `sub_` / `div_`, which is what torchvision's
`ToTensor` + `Normalize` execute.  Writes tests/golden/preprocess.npz."""
from pathlib import Path

import numpy as np
import torch
from PIL import Image

OUT = Path(__file__).resolve().parent / "preprocess.npz"
MEAN = (0.48145466, 0.4578275, 0.40821073)  # open_clip OPENAI_DATASET_MEAN / STD
STD = (0.26862954, 0.26130258, 0.27577711)


def tv_resized_size(h, w, size):
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)  # (oh, ow)


def tv_center_crop(img, size):
    w, h = img.size
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def transform(arr, size, resize_mode, interp):
    img = Image.fromarray(arr, "RGB")
    resample = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR}[interp]
    if resize_mode == "squash":
        img = img.resize((size, size), resample)
    else:
        oh, ow = tv_resized_size(arr.shape[0], arr.shape[1], size)
        if (oh, ow) != arr.shape[:2]:
            img = img.resize((ow, oh), resample)
        img = tv_center_crop(img, size)
    u8 = np.asarray(img, dtype=np.uint8)
    t = torch.from_numpy(u8.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.as_tensor(MEAN, dtype=torch.float32)
    std = torch.as_tensor(STD, dtype=torch.float32)
    t.sub_(mean[:, None, None]).div_(std[:, None, None])
    return u8, t.numpy()


def procedural_image(h, w):
    """Deterministic (h, w, 3) uint8 test image: hard edges + gradients, no RNG (tests regenerate it)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.int64)
    chans = [((yy * 131 + xx * 71 + c * 29 + (yy // 17) * (xx // 23) * 7) % 256) for c in range(3)]
    return np.stack(chans, axis=-1).astype(np.uint8)


def main():
    rng = np.random.default_rng(7)
    cases = [  # (h, w, S, mode, interp)
        (48, 64, 32, "shortest", "bicubic"),    # landscape, downscale 1.5x
        (75, 50, 32, "shortest", "bicubic"),    # portrait (odd crop offset -> round-half-even)
        (77, 50, 32, "shortest", "bicubic"),
        (20, 27, 32, "shortest", "bicubic"),    # upscale
        (32, 32, 32, "shortest", "bicubic"),    # identity
        (32, 57, 32, "shortest", "bicubic"),    # vertical pass skipped (oh == h)
        (331, 97, 32, "shortest", "bicubic"),   # long filter (scale 3), tall
        (40, 200, 32, "squash", "bicubic"),
        (90, 31, 32, "squash", "bilinear"),
        (64, 48, 32, "shortest", "bilinear"),
        (1, 50, 8, "squash", "bicubic"),        # degenerate
        (500, 375, 224, "shortest", "bicubic"), # ImageNet-like
    ]
    out = {"mean": np.float32(MEAN), "std": np.float32(STD), "n": np.int64(len(cases))}
    for i, (h, w, S, mode, interp) in enumerate(cases):
        # smooth + noise so the filters see structure and saturation (clip8) is exercised
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 120 * np.sin(yy[..., None] / 3.1 + np.arange(3)) * np.cos(xx[..., None] / 4.7)
        arr = np.clip(base + rng.normal(0, 60, (h, w, 3)), 0, 255).astype(np.uint8)
        if i % 3 == 0:
            arr[rng.random((h, w)) < 0.2] = (255, 0, 255)  # hard edges -> overshoot beyond [0, 255]
        u8, f = transform(arr, S, mode, interp)
        out[f"img{i}"] = arr
        out[f"cfg{i}"] = np.array([S, {"shortest": 0, "squash": 1}[mode], {"bicubic": 0, "bilinear": 1}[interp]], np.int64)
        if S <= 64:
            out[f"u8_{i}"] = u8
            out[f"f32_{i}"] = f
        else:  # keep the fixture small: checksum + a strided sample of the big case
            out[f"u8_{i}_sum"] = np.int64(u8.astype(np.int64).sum())
            out[f"u8_{i}_sample"] = u8[::7, ::5].copy()
            out[f"f32_{i}_sample"] = f[:, ::7, ::5].copy()
    # large sources, generated procedurally so the fixture stores no input: long filters (ksize up to 127) and a tile
    # of the output as the expectation
    for j, (h, w, S, mode, interp) in enumerate([(1500, 1000, 32, "shortest", "bicubic"), (37, 2900, 48, "squash", "bilinear"),
                                                  (2000, 3000, 224, "shortest", "bicubic")]):
        arr = procedural_image(h, w)
        u8, f = transform(arr, S, mode, interp)
        out[f"proc_cfg{j}"] = np.array([h, w, S, {"shortest": 0, "squash": 1}[mode], {"bicubic": 0, "bilinear": 1}[interp]], np.int64)
        out[f"proc_u8_{j}_sum"] = np.int64(u8.astype(np.int64).sum())
        out[f"proc_u8_{j}_sample"] = u8[::3, ::3].copy()
        out[f"proc_f32_{j}_sample"] = f[:, ::5, ::5].copy()
    out["n_proc"] = np.int64(3)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
