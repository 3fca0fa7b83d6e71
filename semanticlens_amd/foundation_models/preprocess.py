"""Image preprocessing on the device (K12).

The reference hands every sample to the open_clip transform on the host (``foundation_models/clip.py:157-163``:
``torch.stack([self.preprocessor(image) for image in img])``): Pillow resize (antialiased bicubic) -> centre crop ->
``ToTensor`` -> ``Normalize``, one image at a time.  ``DevicePreprocess`` takes the same list of images, uploads the raw
RGB bytes once and produces the identical ``(B, 3, S, S)`` fp32 batch with ``sl_preprocess`` (bit-exact against
Pillow + torchvision arithmetic, ``tests/golden/preprocess.npz``).  No CPU fallback: without a HIP device it raises.
"""
from __future__ import annotations

import numpy as np
import torch

from semanticlens_amd import _native as N

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def _rgb_bytes(img) -> np.ndarray:
    """One sample -> contiguous (h, w, 3) uint8.  PIL ``RGB``/``L`` images, uint8 arrays/tensors (h, w, 3) or (h, w).
    ``L``/2-D inputs are replicated to three channels (resampling commutes with the replication open_clip's
    ``_convert_to_rgb`` performs after the crop).  Other PIL modes resample differently upstream (palette images use
    NEAREST, alpha is premultiplied) and are rejected rather than silently converted."""
    if isinstance(img, torch.Tensor):
        arr = img.detach().cpu().numpy()
    elif isinstance(img, np.ndarray):
        arr = img
    elif hasattr(img, "mode") and hasattr(img, "size"):  # PIL.Image without importing PIL
        if img.mode not in ("RGB", "L"):
            raise ValueError(f"DevicePreprocess supports PIL modes 'RGB' and 'L', got {img.mode!r}: call .convert('RGB') first")
        arr = np.asarray(img)
    else:
        raise TypeError(f"cannot preprocess an object of type {type(img).__name__}")
    if arr.dtype != np.uint8:
        raise TypeError(f"raw images must be uint8, got {arr.dtype}")
    if arr.ndim == 2:
        arr = np.repeat(arr[:, :, None], 3, axis=2)
    if arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError(f"raw image must have shape (h, w, 3) or (h, w), got {arr.shape}")
    return np.ascontiguousarray(arr)


class DevicePreprocess:
    """Callable replacing the open_clip inference transform for whole batches.

    ``size``/``mean``/``std``/``resize_mode``/``interpolation`` carry the meaning of open_clip's ``PreprocessCfg``
    (``resize_mode``: ``"shortest"`` = ``Resize(size)`` + ``CenterCrop(size)``, ``"squash"`` = ``Resize((size, size))``).
    """

    def __init__(self, size: int = 224, mean=OPENAI_DATASET_MEAN, std=OPENAI_DATASET_STD, resize_mode: str = "shortest",
                 interpolation: str = "bicubic", device=None):
        if resize_mode not in N.PP_RESIZE_MODES:
            raise ValueError(f"resize_mode must be one of {sorted(N.PP_RESIZE_MODES)}")
        if interpolation not in N.PP_INTERP:
            raise ValueError(f"interpolation must be one of {sorted(N.PP_INTERP)}")
        self.size = int(size)
        self.mean = tuple(float(v) for v in mean)
        self.std = tuple(float(v) for v in std)
        if len(self.mean) != 3 or len(self.std) != 3:
            raise ValueError("mean and std must have three entries")
        self.resize_mode = resize_mode
        self.interpolation = interpolation
        self._device = torch.device(device) if device is not None else None

    @classmethod
    def from_transform(cls, transform, device=None) -> "DevicePreprocess":
        """Read size / mean / std / interpolation out of a torchvision ``Compose`` as open_clip builds it."""
        size, mean, std, mode, interp = None, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD, "shortest", "bicubic"
        for t in getattr(transform, "transforms", []):
            name = type(t).__name__
            if name == "Resize":
                if isinstance(t.size, (tuple, list)) and len(t.size) == 2:
                    if t.size[0] != t.size[1]:
                        raise ValueError("non-square target sizes are not supported")
                    size, mode = int(t.size[0]), "squash"
                else:
                    size = int(t.size[0] if isinstance(t.size, (tuple, list)) else t.size)
                interp = str(getattr(t.interpolation, "value", t.interpolation)).lower()
            elif name == "CenterCrop":
                cs = t.size[0] if isinstance(t.size, (tuple, list)) else t.size
                if size is not None and int(cs) != size:
                    raise ValueError("Resize and CenterCrop sizes differ: not the open_clip inference transform")
            elif name == "Normalize":
                mean, std = tuple(float(v) for v in t.mean), tuple(float(v) for v in t.std)
        if size is None:
            raise ValueError("no Resize step found in the transform")
        return cls(size, mean, std, mode, interp, device)

    def to(self, device):
        self._device = torch.device(device)
        return self

    def pack(self, imgs):
        """Host side of a batch: raw bytes concatenated into one pinned staging buffer + the launch plan.  Two
        staging buffers alternate; a buffer is reused only after the upload that last read it has completed."""
        arrays = [_rgb_bytes(i) for i in imgs]
        hw = [a.shape[:2] for a in arrays]
        plan, info = N.preprocess_plan(hw, self.size, self.resize_mode, self.interpolation)
        total = info["pixel_bytes"]
        slot = self._slot = (getattr(self, "_slot", 1) + 1) % 2
        stage = self.__dict__.setdefault("_staging", [None, None])
        events = self.__dict__.setdefault("_events", [None, None])
        if events[slot] is not None:
            events[slot].synchronize()
            events[slot] = None
        if stage[slot] is None or stage[slot].numel() < max(total, 1):
            buf = torch.empty((max(total, 1) * 5 // 4 + 64,), dtype=torch.uint8)
            stage[slot] = buf.pin_memory() if torch.cuda.is_available() else buf
        buf = stage[slot][: max(total, 1)]
        if arrays:
            np.concatenate([a.reshape(-1) for a in arrays], out=buf.numpy()[:total])
        return buf, plan, info

    def __call__(self, img):
        single = not isinstance(img, (list, tuple))
        if isinstance(img, torch.Tensor) and img.ndim == 4:  # (B, h, w, 3) uint8 batch, possibly already on the device
            return self._uniform_batch(img)
        imgs = [img] if single else list(img)
        dev = self._device or N.default_device()
        buf, plan, info = self.pack(imgs)
        pixels = buf.to(dev, non_blocking=True)
        if buf.is_pinned():  # the staging buffer may be rewritten once this upload is done
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._events[self._slot] = ev
        out, _ = N.preprocess(pixels, plan, info, self.size, self.mean, self.std, self.interpolation)
        return out[0] if single else out

    def _uniform_batch(self, batch: torch.Tensor):
        if batch.dtype != torch.uint8 or batch.shape[-1] != 3:
            raise TypeError("a tensor batch must be (B, h, w, 3) uint8")
        dev = batch.device if batch.is_cuda else (self._device or N.default_device())
        B, h, w, _ = batch.shape
        plan, info = N.preprocess_plan([(h, w)] * B, self.size, self.resize_mode, self.interpolation)
        pixels = batch.to(dev, non_blocking=True).contiguous().reshape(-1)
        out, _ = N.preprocess(pixels, plan, info, self.size, self.mean, self.std, self.interpolation)
        return out
