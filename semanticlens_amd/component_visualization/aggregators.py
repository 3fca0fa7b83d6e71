"""Aggregators: reduce a layer activation to one value per (sample, component).

Mirror of ``semanticlens/component_visualization/aggregators.py`` (reference v0.2.1): same
function names (they are part of the cache file names, aggregators.py:27,32), same argument
checks and error text, same ``(B, C)`` result.  The arithmetic is the HIP reduce kernel
(``csrc/reduce.hip``: K1 for conv maps, K2 for token sequences) instead of
``tensor.clone()...amax/mean(...).cpu()``.

Each function carries ``_sl_native = (kind, code, pos)``.  ``ActMaxCache`` uses it to run the
fused path — reduce + bf16 cast + top-k merge on the device with no host round trip — instead
of calling the function and shipping a ``(B, C)`` tensor through the host.
"""
from __future__ import annotations

import torch

from semanticlens_amd import _native as N

_ERROR_MESSAGE = f"(Select or implement a different aggregation function in {__file__}.)"


def _reduce(tensor: torch.Tensor, kind: str, code: int, pos: int = 0) -> torch.Tensor:
    """Run K1/K2 and return the (B, C) result like the reference does: activation dtype, on the host."""
    x = N.to_device(tensor.detach())
    out = torch.empty((x.shape[0], x.shape[1] if kind == "conv" else x.shape[2]), dtype=torch.float32, device=x.device)
    if kind == "conv":
        N.reduce_conv(x, code, None, out)
    else:
        N.reduce_tokens(x, code, pos, None, out)
    return out.to(tensor.dtype).cpu()


def _native(kind: str, code: int, pos: int = 0):
    def mark(fn):
        fn._sl_native = (kind, code, pos)
        return fn

    return mark


@_native("conv", N.SL_CONV_MEAN)
def aggregate_conv_mean(tensor: torch.Tensor) -> torch.Tensor:
    """Mean over H*W of a (B, C, H, W) activation -> (B, C).  Reference: aggregators.py:38-61."""
    if tensor.ndim != 4:
        raise ValueError("Input tensor should be 4D. \n" + _ERROR_MESSAGE)
    if isinstance(tensor, tuple):  # unreachable for tuples (``.ndim`` fails first), as in the reference
        tensor = tensor[0]
    return _reduce(tensor, "conv", N.SL_CONV_MEAN)


@_native("conv", N.SL_CONV_MAX)
def aggregate_conv_max(tensor: torch.Tensor) -> torch.Tensor:
    """Max over H*W of a (B, C, H, W) activation -> (B, C).  Reference: aggregators.py:64-87."""
    if tensor.ndim != 4:
        raise ValueError("Input tensor should be 4D. \n" + _ERROR_MESSAGE)
    if isinstance(tensor, tuple):
        tensor = tensor[0]
    return _reduce(tensor, "conv", N.SL_CONV_MAX)


def _check_3d(tensor):
    if tensor.ndim != 3:
        raise ValueError("Input tensor should be 3D. \n" + _ERROR_MESSAGE)
    if isinstance(tensor, tuple):
        tensor = tensor[0]
    return tensor


@_native("tokens", N.SL_TOK_MEAN)
def aggregate_transformer_mean(tensor: torch.Tensor) -> torch.Tensor:
    """Mean over tokens of a (B, T, F) activation -> (B, F).  Reference: aggregators.py:90-114."""
    return _reduce(_check_3d(tensor), "tokens", N.SL_TOK_MEAN)


@_native("tokens", N.SL_TOK_ABSMEAN)
def aggregate_transformer_absmean(tensor: torch.Tensor) -> torch.Tensor:
    """Mean of |x| over tokens -> (B, F).  Reference: aggregators.py:117-141."""
    return _reduce(_check_3d(tensor), "tokens", N.SL_TOK_ABSMEAN)


@_native("tokens", N.SL_TOK_MAX)
def aggregate_transformer_max(tensor: torch.Tensor) -> torch.Tensor:
    """Max over tokens -> (B, F).  Reference: aggregators.py:144-168."""
    return _reduce(_check_3d(tensor), "tokens", N.SL_TOK_MAX)


@_native("tokens", N.SL_TOK_ABSMAX)
def aggregate_transformer_absmax(tensor: torch.Tensor) -> torch.Tensor:
    """Max of |x| over tokens -> (B, F).  Reference: aggregators.py:171-195."""
    return _reduce(_check_3d(tensor), "tokens", N.SL_TOK_ABSMAX)


def get_aggregate_transformer_special_token(token_position: int):
    """Factory: pick the activation at one token position (e.g. 0 = CLS).  Reference: aggregators.py:198-244."""

    @_native("tokens", N.SL_TOK_TOKEN, int(token_position))
    def aggregate_transformer_special_token(tensor: torch.Tensor) -> torch.Tensor:
        tensor = _check_3d(tensor)
        T = tensor.shape[1]
        if not -T <= token_position < T:
            raise IndexError(f"index {token_position} is out of bounds for dimension 1 with size {T}")
        return _reduce(tensor, "tokens", N.SL_TOK_TOKEN, int(token_position))

    return aggregate_transformer_special_token
