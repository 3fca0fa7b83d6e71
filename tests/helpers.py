"""Deterministic fakes shared by tests/golden/make_golden.py (reference side) and the tests (build side).

Everything here is plain torch/numpy so that the reference and the build can be
driven with *identical* objects.  Values are chosen to be exactly representable
(small integers / dyadic rationals) wherever a test demands bit-exact results
across CPU and GPU arithmetic orders.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn


class FakeVLM:
    """A deterministic stand-in for an ``AbstractVLM`` (foundation_models/base.py:12-120).

    * ``tokenize``: strings -> (n, L) int64 of character codes (0-padded).
    * ``encode_text``: histogram of character codes -> fixed integer projection.
    * ``preprocess``: list of tensors (or one tensor) -> stacked float tensor.
    * ``encode_image``: flatten -> fixed integer projection.
    All weights are small integers, so outputs are exact in fp32 on any device.
    """

    name = "fake-vlm"

    def __init__(self, dim: int = 16, img_numel: int = 3 * 16 * 16, ctx: int = 24, seed: int = 7):
        g = np.random.RandomState(seed)
        self.dim, self.ctx = dim, ctx
        self.w_txt = torch.from_numpy(g.randint(-3, 4, size=(128, dim)).astype(np.float32))
        self.w_img = torch.from_numpy(g.randint(-2, 3, size=(img_numel, dim)).astype(np.float32))
        self._device = torch.device("cpu")
        self.calls = {"encode_text": 0, "encode_image": 0}

    @property
    def device(self):
        return self._device

    def to(self, device):
        self._device = torch.device(device)
        self.w_txt = self.w_txt.to(self._device)
        self.w_img = self.w_img.to(self._device)
        return self

    def tokenize(self, txt):
        if isinstance(txt, str):
            txt = [txt]
        out = torch.zeros(len(txt), self.ctx, dtype=torch.int64)
        for i, s in enumerate(txt):
            codes = [min(ord(ch), 127) for ch in s][: self.ctx]
            out[i, : len(codes)] = torch.tensor(codes, dtype=torch.int64)
        return out.to(self._device)

    def encode_text(self, tokens):
        self.calls["encode_text"] += 1
        onehot = torch.nn.functional.one_hot(tokens, 128).to(torch.float32)
        onehot[..., 0] = 0  # padding
        return onehot.sum(1) @ self.w_txt

    def preprocess(self, img):
        if isinstance(img, (list, tuple)):
            x = torch.stack([torch.as_tensor(i) for i in img])
        else:
            x = torch.as_tensor(img)
            if x.ndim == 3:
                x = x.unsqueeze(0)
        return x.to(torch.float32).to(self._device)

    def encode_image(self, x):
        self.calls["encode_image"] += 1
        return x.flatten(1).to(torch.float32) @ self.w_img


def make_int_conv_model(seed: int = 3) -> nn.Module:
    """Conv stack whose weights/biases are small integers: activations on integer
    inputs are integers (< 2**24), hence identical under any summation order."""
    g = np.random.RandomState(seed)
    m = nn.Sequential(nn.Conv2d(3, 8, 3), nn.ReLU(), nn.Conv2d(8, 16, 3))
    with torch.no_grad():
        for mod in m:
            if isinstance(mod, nn.Conv2d):
                mod.weight.copy_(torch.from_numpy(g.randint(-2, 3, size=tuple(mod.weight.shape)).astype(np.float32)))
                mod.bias.copy_(torch.from_numpy(g.randint(-3, 4, size=tuple(mod.bias.shape)).astype(np.float32)))
    m.name = "int-conv"
    return m.eval()


def make_int_images(n: int, seed: int = 5, hw: int = 16) -> torch.Tensor:
    g = np.random.RandomState(seed)
    return torch.from_numpy(g.randint(-4, 5, size=(n, 3, hw, hw)).astype(np.float32))


class TensorPairDataset(torch.utils.data.Dataset):
    """``(tensor, label)`` items like the reference's ``dataset_model`` (activation_based.py:351)."""

    def __init__(self, x: torch.Tensor, name: str = "int-images"):
        self.x = x
        self.name = name

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], 0


def tie_free_bf16_matrix(n: int, c: int, seed: int, frac_negative: float = 0.25) -> np.ndarray:
    """(n, c) fp32, every column holds n *distinct* bf16-representable values
    (so fp32->bf16 is exact and top-k has no ties); a fraction is negative."""
    g = np.random.RandomState(seed)
    out = np.empty((n, c), dtype=np.float32)
    lo, hi = 0x3000, 0x4800  # positive normal bf16 patterns ~[4.7e-10, 1.3e5]
    assert n <= hi - lo
    for j in range(c):
        pats = g.choice(np.arange(lo, hi, dtype=np.uint32), size=n, replace=False)
        neg = g.rand(n) < frac_negative
        pats = pats | (neg.astype(np.uint32) << 15)
        out[:, j] = (pats << 16).view(np.float32)
    return out
