"""Naming helper used by the cache paths (reference: utils/helper.py:11-18)."""
from __future__ import annotations

import hashlib


def get_fallback_name(obj) -> str:
    """``<ClassName>-<int(sha256(str(obj)))>`` — identical to the reference so cache directories interchange."""
    return obj.__class__.__name__ + "-" + str(int(hashlib.sha256(str(obj).encode()).hexdigest(), 16))


class _Denormalize:
    """``x * std + mean`` per channel, evaluated the way the reference's two chained ``Normalize`` steps do
    (utils/helper.py:36-73): ``(x - 0) / (1 / std)`` then ``(y - (-mean)) / 1``.  Plain callable (torchvision, which
    the reference builds this from, is not a dependency here)."""

    def __init__(self, mean, std):
        import torch

        self.mean = torch.as_tensor(mean, dtype=torch.float32).reshape(-1)
        self.std = torch.as_tensor(std, dtype=torch.float32).reshape(-1)

    def __call__(self, x):
        shape = (-1, 1, 1) if x.ndim == 3 else (1, -1, 1, 1)
        inv = (1 / self.std).to(x.device).reshape(shape)
        neg_mean = (-self.mean).to(x.device).reshape(shape)
        return x.sub(0.0).div(inv).sub(neg_mean).div(1.0)

    def __repr__(self):
        return f"Denormalize(mean={self.mean.tolist()}, std={self.std.tolist()})"


def get_denormalization_transform(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """Transform undoing a per-channel normalisation for display (reference: utils/helper.py:36-73; defaults are the
    ImageNet statistics).  Accepts ``(C, H, W)`` or ``(B, C, H, W)`` tensors."""
    return _Denormalize(mean, std)
