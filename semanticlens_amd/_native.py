"""ctypes binding of ``libsemanticlens_hip.so`` (C ABI: ``include/semanticlens_amd.h``).

This is the only module that talks to the native library.  There is NO CPU
fallback: if the library is missing, or a tensor cannot be placed on a HIP
device, the call raises.  PyTorch is used for device memory and streams only
(``tensor.data_ptr()`` / ``torch.cuda.current_stream()``).
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libsemanticlens_hip.so"

# enums of include/semanticlens_amd.h
SL_F32, SL_F16, SL_BF16 = 0, 1, 2
SL_CONV_MAX, SL_CONV_MEAN, SL_CONV_SUM = 0, 1, 2
SL_TOK_MEAN, SL_TOK_ABSMEAN, SL_TOK_MAX, SL_TOK_ABSMAX, SL_TOK_TOKEN = 0, 1, 2, 3, 4
SL_TIES_TOTAL, SL_TIES_ATEN = 0, 1
SL_MAX_SLOTS = 16
SL_PROF_REDUCE, SL_PROF_MERGE, SL_PROF_GEMM, SL_PROF_GATHER, SL_PROF_SCORES = 0, 1, 2, 3, 4
SL_ACT_NONE, SL_ACT_GELU, SL_ACT_QUICKGELU, SL_ACT_GELU_TANH = 0, 1, 2, 3
TIE_MODES = {"total": SL_TIES_TOTAL, "aten": SL_TIES_ATEN}
SL_PP_PLAN_STRIDE = 16
PP_RESIZE_MODES = {"shortest": 0, "squash": 1}
PP_INTERP = {"bicubic": 0, "bilinear": 1}

_DTYPES = {torch.float32: SL_F32, torch.float16: SL_F16, torch.bfloat16: SL_BF16}

_i64 = ctypes.c_int64
_vp = ctypes.c_void_p
_int = ctypes.c_int
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/semanticlens_amd.h declares
SIGNATURES = {
    "sl_last_error": (ctypes.c_char_p, []),
    "sl_abi_version": (_int, []),
    "sl_device_count": (_int, []),
    "sl_reduce_conv": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp]),
    "sl_reduce_tokens": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _int, _i64, _vp, _vp, _vp]),
    "sl_actmax_init": (_int, [_vp, _vp, _i64, _i64, _vp]),
    "sl_actmax_merge": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _int, _vp]),
    "sl_actmax_update": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _int, _vp, _sz, _vp]),
    "sl_actmax_aten_ws_bytes": (_sz, [_i64, _i64, _i64]),
    "sl_actmax_update_multi": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _vp]),
    "sl_actmax_update_multi_supported": (_int, [_i64, _i64, _i64]),
    "sl_aten_topk_order_host": (_int, [_vp, _i64, _i64, _vp]),
    "sl_set_option": (_int, [ctypes.c_char_p, _i64]),
    "sl_get_option": (_i64, [ctypes.c_char_p]),
    "sl_reduce_conv_multi": (_int, [_vp, _int, _int, _i64, _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp]),
    "sl_reduce_tokens_multi": (_int, [_vp, _int, _int, _i64, _i64, _i64, _i64, _i64, _i64, _int, _i64, _vp, _vp]),
    "sl_actmax_merge_states": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    "sl_comm_unique_id": (_int, [_vp]),
    "sl_comm_init_from_unique_id": (_int, [_vp, _int, _int, ctypes.POINTER(_vp)]),
    "sl_comm_destroy": (_int, [_vp]),
    "sl_comm_info": (_int, [_vp, ctypes.POINTER(_int), ctypes.POINTER(_int), ctypes.POINTER(_int)]),
    "sl_comm_allgather": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "sl_comm_allreduce": (_int, [_vp, _vp, _i64, _int, _int, _vp]),
    "sl_actmax_packed_bytes": (_sz, [_int, _vp, _i64]),
    "sl_actmax_pack": (_int, [_int, _vp, _vp, _vp, _i64, _vp, _vp]),
    "sl_actmax_merge_packed": (_int, [_int, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "sl_actmax_allgather_merge_ws_bytes": (_sz, [_int, _vp, _i64, _int]),
    "sl_actmax_allgather_merge": (_int, [_vp, _int, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "sl_gather_rows": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp]),
    "sl_gather_rows_shard": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "sl_similarity": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "sl_similarity_ws_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "sl_set_gemm_mode": (_int, [_int]),
    "sl_set_reduce_policy": (_int, [_i64, _i64]),
    "sl_abs_norm_rows": (_int, [_vp, _i64, _i64, ctypes.c_float, _vp]),
    "sl_similarity_multi": (_int, [_vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _sz, _vp]),
    "sl_similarity_multi_ws_bytes": (_sz, [_i64, _i64, _vp, _int]),
    "sl_clarity": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "sl_clarity_multi": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _vp]),
    "sl_redundancy": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "sl_redundancy_ws_bytes": (_sz, [_i64, _i64, _i64]),
    "sl_template_mean": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "sl_poly2means": (_int, [_vp, _i64, _i64, _i64, _vp, _int, _vp, _int, _vp, _vp, _vp, _sz, _vp]),
    "sl_poly2means_ws_bytes": (_sz, [_i64, _i64, _i64]),
    "sl_kmeans_trials": (_int, [_int]),
    "sl_polykmeans": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _int, _vp, _int, _vp, _vp, _vp, _sz, _vp]),
    "sl_polykmeans_ws_bytes": (_sz, [_i64, _i64, _i64, _int, _int]),
    "sl_linear": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _int, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "sl_layernorm": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, ctypes.c_float, _vp, _vp, _i64, _vp]),
    "sl_attention": (_int, [_vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp]),
    "sl_attention_bf16x3": (_int, [_vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp]),
    "sl_patchify": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "sl_attention_pool": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "sl_attention_pool_q": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "sl_tokens_from_map": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "sl_split_elems": (_sz, [_i64, _i64]),
    "sl_split_bf16": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "sl_linear_bf16x3": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _int, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "sl_broadcast_row": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "sl_embed_tokens": (_int, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "sl_preprocess_plan": (_int, [_vp, _vp, _i64, _int, _int, _int, _vp, _vp]),
    "sl_preprocess": (_int, [_vp, _vp, _i64, _int, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sl_prof_enable": (_int, [_int]),
    "sl_prof_reset": (_int, []),
    "sl_prof_read": (_int, [_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    """The HIP library is missing or a native call failed."""


def lib():
    """Load (once) and return the native library; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("SEMANTICLENS_AMD_LIB", LIB_PATH))
        if not path.exists():
            raise NativeLibraryError(
                f"{path} not found. semanticlens_amd has no CPU fallback: build the HIP library first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C semanticlens_amd/csrc)."
            )
        handle = ctypes.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _check(rc: int, what: str) -> int:
    if rc < 0:
        msg = lib().sl_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg or f"{what}: invalid argument")
        raise NativeLibraryError(f"{what} failed ({rc}): {msg}")
    return rc


def default_device() -> torch.device:
    """The HIP device native state lives on.  Raises when there is none (no CPU fallback)."""
    if not torch.cuda.is_available():
        raise NativeLibraryError(
            "semanticlens_amd needs a HIP device (MI355X): torch.cuda.is_available() is False and there is "
            "no CPU fallback for the concept-DB hot path."
        )
    return torch.device("cuda", torch.cuda.current_device())


def resolve_device(device=None) -> torch.device:
    """``device`` with its index filled in: ``"cuda"`` means the CURRENT device (the usual pattern after
    ``torch.cuda.set_device(rank)``), so that ``tensor.device == resolve_device(...)`` holds for tensors living there."""
    if device is None:
        return default_device()
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        return default_device()
    return dev


def to_device(t: torch.Tensor, device: torch.device | None = None) -> torch.Tensor:
    """Place ``t`` on a HIP device (host->device copies are plumbing, not compute)."""
    if t.is_cuda:
        return t
    return t.to(device or default_device())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor):
    """torch's current stream on the tensor's device as a raw ``hipStream_t`` (every kernel is enqueued on it)."""
    if _raw_stream is not None:  # 0.2 us; the Stream object route costs 1.8 of a call's ~9 us on the host
        return _vp(_raw_stream(t.device.index))
    return _vp(torch.cuda.current_stream(t.device).cuda_stream)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(device: torch.device):
    """Device guard for a native call: nothing to do when ``device`` is already current (the usual case; saves 1.2 us)."""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


def _ptr(t: torch.Tensor | None):
    return _vp(t.data_ptr()) if t is not None and t.numel() > 0 else _vp(None)


def _need_f32(what: str, *tensors):
    """The encoder entry points take raw fp32 device pointers: refuse anything else instead of reinterpreting its bytes."""
    for t in tensors:
        if t is not None and (t.dtype != torch.float32 or not t.is_cuda):
            raise TypeError(f"{what}: expected float32 tensors on a HIP device, got {t.dtype} on {t.device}")


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f"activation dtype {t.dtype} is not supported (float32, float16, bfloat16)") from None


# ------------------------------------------------------------------------------------------------
# K1 / K2
# ------------------------------------------------------------------------------------------------
def _flatten_spatial(x: torch.Tensor):
    """(B,C,H,W) -> strides (sb, sc, ss) of the (B,C,H*W) view, making a copy only if H,W cannot merge.  The stride of a
    size-1 dimension carries no information (torch keeps whatever the view had), so it is never used."""
    B, C, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if H == 1:
        return x, sb, sc, sw
    if W == 1:
        return x, sb, sc, sh
    if sh != W * sw:
        x = x.contiguous()
        return x, C * H * W, H * W, 1
    return x, sb, sc, sw


def reduce_conv(x: torch.Tensor, agg: int, cand: torch.Tensor | None, out_f32: torch.Tensor | None):
    """Launch K1 on a 4-D device tensor.  ``cand`` (B,C) bf16 and/or ``out_f32`` (B,C) f32 are filled."""
    assert x.is_cuda and x.ndim == 4
    B, C, H, W = x.shape
    x, sb, sc, ss = _flatten_spatial(x)
    with _on(x.device):
        rc = lib().sl_reduce_conv(_ptr(x), _dtype_code(x), B, C, H * W, sb, sc, ss, agg, _ptr(cand), _ptr(out_f32), _stream(x))
    _check(rc, "sl_reduce_conv")


def abs_norm_rows(x: torch.Tensor, eps: float = 1e-10) -> torch.Tensor:
    """In place on a contiguous (B, C) fp32 device tensor: ``x[b] /= x[b].abs().sum() + eps`` (the relevance visualizer's
    ``abs_norm``)."""
    assert x.is_cuda and x.ndim == 2 and x.dtype == torch.float32 and x.is_contiguous()
    with _on(x.device):
        _check(lib().sl_abs_norm_rows(_ptr(x), x.shape[0], x.shape[1], float(eps), _stream(x)), "sl_abs_norm_rows")
    return x


def reduce_tokens(x: torch.Tensor, agg: int, pos: int, cand: torch.Tensor | None, out_f32: torch.Tensor | None):
    """Launch K2 on a 3-D device tensor (B,T,F)."""
    assert x.is_cuda and x.ndim == 3
    B, T, F = x.shape
    sb, st, sf = x.stride()
    with _on(x.device):
        rc = lib().sl_reduce_tokens(
            _ptr(x), _dtype_code(x), B, T, F, sb, st, sf, agg, pos, _ptr(cand), _ptr(out_f32), _stream(x)
        )
    _check(rc, "sl_reduce_tokens")


def reduce_multi(kind: str, xs: list[torch.Tensor], agg: int, pos: int, cand: torch.Tensor):
    """K1 / K2 over ``L`` device tensors of ONE shape, strides and dtype into ``cand`` ``(L, B, C)`` bf16: one launch when the
    component axis is contiguous (``sl_reduce_*_multi``)."""
    x0 = xs[0]
    L = len(xs)
    assert all(x.is_cuda and x.shape == x0.shape and x.stride() == x0.stride() and x.dtype == x0.dtype for x in xs)
    assert cand.is_contiguous() and cand.dtype == torch.bfloat16 and cand.shape[0] == L
    if kind == "conv":
        flat = [_flatten_spatial(x) for x in xs]  # identical strides: the same answer (view or copy) for every tensor
        xs = [f[0] for f in flat]
        sb, sc, ss = flat[0][1:]
    ptrs = (_vp * L)(*[x.data_ptr() for x in xs])
    with _on(x0.device):
        if kind == "conv":
            B, C = x0.shape[:2]
            S = x0.shape[2] * x0.shape[3]
            rc = lib().sl_reduce_conv_multi(ptrs, L, _dtype_code(x0), B, C, S, sb, sc, ss, agg, _ptr(cand), _stream(x0))
            _check(rc, "sl_reduce_conv_multi")
        else:
            B, T, F = x0.shape
            sb, st, sf = x0.stride()
            rc = lib().sl_reduce_tokens_multi(ptrs, L, _dtype_code(x0), B, T, F, sb, st, sf, agg, pos, _ptr(cand), _stream(x0))
            _check(rc, "sl_reduce_tokens_multi")


# ------------------------------------------------------------------------------------------------
# K3 / K4
# ------------------------------------------------------------------------------------------------
def actmax_update_multi_supported(C: int, k: int, B: int) -> bool:
    return bool(lib().sl_actmax_update_multi_supported(C, k, B))


def actmax_update_multi(states: list[tuple[torch.Tensor, torch.Tensor]], cands: list[torch.Tensor], id_bases: list[int], B: int):
    """``SL_TIES_ATEN`` update of ``L`` states ``(C_l, k)`` from ``L`` candidate matrices ``(B, C_l)`` bf16 in one launch."""
    L = len(states)
    k = states[0][0].shape[1]
    assert len(cands) == L and all(c.is_contiguous() and tuple(c.shape) == (B, v.shape[0]) and v.shape[1] == k for c, (v, _) in zip(cands, states))
    vp = (_vp * L)(*[v.data_ptr() for v, _ in states])
    ip = (_vp * L)(*[i.data_ptr() for _, i in states])
    cp = (_vp * L)(*[c.data_ptr() for c in cands])
    hb = (_i64 * L)(*id_bases)
    hc = (_i64 * L)(*[v.shape[0] for v, _ in states])
    with _on(cands[0].device):
        rc = lib().sl_actmax_update_multi(vp, ip, hb, hc, cp, L, k, B, _stream(cands[0]))
    _check(rc, "sl_actmax_update_multi")


def actmax_init(vals: torch.Tensor, ids: torch.Tensor):
    C, k = vals.shape
    with _on(vals.device):
        _check(lib().sl_actmax_init(_ptr(vals), _ptr(ids), C, k, _stream(vals)), "sl_actmax_init")


def actmax_merge(vals, ids, cand: torch.Tensor, slot_stride: int, id_bases: list[int], rows: list[int]):
    C, k = vals.shape
    n = len(rows)
    hb = (_i64 * max(n, 1))(*id_bases)
    hr = (_i64 * max(n, 1))(*rows)
    with _on(vals.device):
        rc = lib().sl_actmax_merge(_ptr(vals), _ptr(ids), C, k, _ptr(cand), slot_stride, hb, hr, n, _stream(vals))
    _check(rc, "sl_actmax_merge")


def set_option(name: str, value: int) -> None:
    """Force one of several bit-identical kernel variants (``g3_tile``, ``f32_tile``, ``g3_strip_off``, ``colreduce_nw``; 0 = the
    dispatcher's own rule).  For the parity tests; ``SL_OPTIONS="name=value,..."`` presets them for a process."""
    _check(lib().sl_set_option(name.encode(), int(value)), "sl_set_option")


def get_option(name: str) -> int:
    return int(lib().sl_get_option(name.encode()))


def aten_topk_order_host(row_bf16: torch.Tensor, k: int) -> torch.Tensor:
    """Positions ``torch.topk(row, k)`` selects on the CPU according to the library's restatement (host code, no device)."""
    row = row_bf16.detach().to("cpu", torch.bfloat16).contiguous()
    out = torch.empty(k, dtype=torch.int32)
    _check(lib().sl_aten_topk_order_host(row.data_ptr(), row.numel(), k, out.data_ptr()), "sl_aten_topk_order_host")
    return out


_ATEN_SELFTEST: dict | None = None


def aten_order_selftest(rows: int = 200, force: bool = False) -> dict:
    """Once per process (first ``tie_mode="aten"`` use): the restatement the K3 kernels evaluate against the INSTALLED
    ``torch.topk`` on ``rows`` tie-heavy bf16 rows covering both of ATen's branches (``k * 64 <= n``: partial_sort; else
    nth_element + sort) — ~1 ms of host work.  The restatement is pinned to libstdc++ 11's introselect / introsort and
    torch 2.10's TopKImpl.h (the reference locks torch 2.7.1, same code); a host whose pair orders ties differently gets a
    warning naming the versions, because ``aten`` ids would then differ from what ``torch.topk`` gives THERE."""
    global _ATEN_SELFTEST
    if _ATEN_SELFTEST is not None and not force:
        return _ATEN_SELFTEST
    g = torch.Generator().manual_seed(1234)
    bad, cases = 0, 0
    shapes = [(20, 64), (20, 256), (100, 32), (5, 7), (1, 1), (3, 250), (2, 128), (20, 2000), (100, 6500)]  # (k, B): n = k + B
    for i in range(rows):
        k, B = shapes[i % len(shapes)]
        n = k + B
        row = (torch.randint(0, 12, (n,), generator=g).float() / 4).to(torch.bfloat16)  # ties are the point
        if i % 7 == 0:
            row[int(torch.randint(0, n, (1,), generator=g))] = float("nan")
        if i % 5 == 0:
            row[:k] = -0.0  # a fresh state: the sentinel (activation_caching.py:108)
        want = torch.topk(row, k).indices.to(torch.int32)
        got = aten_topk_order_host(row, k)
        cases += 1
        bad += int(not torch.equal(want, got))
    _ATEN_SELFTEST = {"rows": cases, "mismatches": bad, "torch": torch.__version__}
    if bad:
        import warnings

        warnings.warn(
            f"semanticlens_amd: tie_mode='aten' reproduces torch.topk's CPU tie order as restated for libstdc++ 11 / torch 2.7-2.10 "
            f"(ATen TopKImpl.h); the installed torch {torch.__version__} selects different positions on {bad} of {cases} tie-heavy rows. "
            "Top-k VALUES are unaffected; sample ids among tied values may differ from torch.topk on this host.  tie_mode='total' is "
            "independent of the host library.", RuntimeWarning, stacklevel=3)
    return _ATEN_SELFTEST


def actmax_aten_ws_bytes(C: int, k: int, B: int) -> int:
    return int(lib().sl_actmax_aten_ws_bytes(C, k, B))


def actmax_update(vals, ids, cand: torch.Tensor, sample_ids: torch.Tensor | None, id_base: int, B: int, ties: int,
                  ws: torch.Tensor | None):
    C, k = vals.shape
    with _on(vals.device):
        rc = lib().sl_actmax_update(
            _ptr(vals), _ptr(ids), C, k, _ptr(cand), _ptr(sample_ids), id_base, B, ties, _ptr(ws),
            ws.numel() * ws.element_size() if ws is not None else 0, _stream(vals),
        )
    _check(rc, "sl_actmax_update")


def actmax_merge_states(vals, ids, other_vals: torch.Tensor, other_ids: torch.Tensor):
    C, k = vals.shape
    R = other_vals.shape[0]
    with _on(vals.device):
        rc = lib().sl_actmax_merge_states(_ptr(vals), _ptr(ids), C, k, _ptr(other_vals), _ptr(other_ids), R, _stream(vals))
    _check(rc, "sl_actmax_merge_states")


# ------------------------------------------------------------------------------------------------
# K4 with its exchange step: RCCL behind the C ABI (csrc/comm.hip)
# ------------------------------------------------------------------------------------------------
SL_COMM_ID_BYTES = 128
_COMM_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int64: 2}
_COMM_OPS = {"sum": 0, "max": 1, "min": 2}


def _state_tables(states):
    """ctypes tables (C[], vals*[], ids*[]) + k of live device states [(vals (C,k) bf16, ids (C,k) int64)]."""
    k = int(states[0][0].shape[1])
    dev = states[0][0].device
    for vals, ids in states:
        if not (vals.is_cuda and ids.is_cuda and vals.device == dev and ids.device == dev and vals.is_contiguous() and ids.is_contiguous()):
            raise ValueError("states are contiguous tensors on one HIP device")
        if vals.dtype != torch.bfloat16 or ids.dtype != torch.int64 or vals.shape != ids.shape or vals.ndim != 2 or vals.shape[1] != k:
            raise ValueError("states are (C, k) bf16 values + (C, k) int64 ids with one k")
    n = len(states)
    C = (_i64 * n)(*[int(v.shape[0]) for v, _ in states])
    pv = (_vp * n)(*[v.data_ptr() for v, _ in states])
    pi = (_vp * n)(*[i.data_ptr() for _, i in states])
    return n, C, pv, pi, k


def actmax_pack(states) -> torch.Tensor:
    """One rank's packed block of all layers' states (uint8, ``sl_actmax_packed_bytes`` long; layout of ``sl_actmax_pack``)."""
    n, C, pv, pi, k = _state_tables(states)
    dev = states[0][0].device
    out = torch.empty(int(lib().sl_actmax_packed_bytes(n, C, k)), dtype=torch.uint8, device=dev)
    with _on(dev):
        rc = lib().sl_actmax_pack(n, pv, pi, C, k, _ptr(out), _stream(out))
    _check(rc, "sl_actmax_pack")
    return out


def actmax_merge_packed(states, gathered: torch.Tensor, skip_rank: int = -1):
    """K4 of every layer against ``gathered`` = ``(R, packed_bytes)`` uint8 blocks (e.g. all-gathered), in place."""
    n, C, pv, pi, k = _state_tables(states)
    dev = states[0][0].device
    P = int(lib().sl_actmax_packed_bytes(n, C, k))
    if not (gathered.is_cuda and gathered.device == dev and gathered.dtype == torch.uint8 and gathered.is_contiguous()
            and gathered.ndim == 2 and gathered.shape[1] == P):
        raise ValueError(f"gathered states are a contiguous (R, {P}) uint8 tensor on {dev}")
    with _on(dev):
        rc = lib().sl_actmax_merge_packed(n, pv, pi, C, k, _ptr(gathered), gathered.shape[0], int(skip_rank), _stream(gathered))
    _check(rc, "sl_actmax_merge_packed")


class Comm:
    """One RCCL communicator of the native library (``sl_comm_*``): one process per GPU, created collectively.

    ``Comm.unique_id()`` on rank 0 -> hand the 128 bytes to every process -> ``Comm(id, world, rank, device)`` on all of
    them.  Collectives are enqueued on torch's current stream of ``device``."""

    def __init__(self, unique_id: bytes, world: int, rank: int, device=None):
        if len(unique_id) != SL_COMM_ID_BYTES:
            raise ValueError(f"an RCCL unique id is {SL_COMM_ID_BYTES} bytes, got {len(unique_id)}")
        self.device = resolve_device(device)
        self.world, self.rank = int(world), int(rank)
        handle = _vp()
        buf = ctypes.create_string_buffer(bytes(unique_id), SL_COMM_ID_BYTES)
        with _on(self.device):
            rc = lib().sl_comm_init_from_unique_id(ctypes.cast(buf, _vp), self.world, self.rank, ctypes.byref(handle))
        _check(rc, "sl_comm_init_from_unique_id")
        self._h = handle

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(SL_COMM_ID_BYTES)
        _check(lib().sl_comm_unique_id(ctypes.cast(buf, _vp)), "sl_comm_unique_id")
        return buf.raw

    def info(self):
        w, r, d = _int(), _int(), _int()
        _check(lib().sl_comm_info(self._h, ctypes.byref(w), ctypes.byref(r), ctypes.byref(d)), "sl_comm_info")
        return w.value, r.value, d.value

    def destroy(self):
        if self._h is not None:
            h, self._h = self._h, None
            _check(lib().sl_comm_destroy(h), "sl_comm_destroy")

    def _dev(self, t: torch.Tensor):
        if not (t.is_cuda and t.device == self.device and t.is_contiguous()):
            raise ValueError(f"collectives take contiguous tensors on {self.device}")
        return t

    def allgather(self, send: torch.Tensor) -> torch.Tensor:
        """``(world,) + send.shape``: block r is rank r's ``send`` (equal sizes on every rank)."""
        send = self._dev(send)
        recv = torch.empty((self.world,) + tuple(send.shape), dtype=send.dtype, device=self.device)
        with _on(self.device):
            rc = lib().sl_comm_allgather(self._h, _ptr(send), _ptr(recv), send.numel() * send.element_size(), _stream(send))
        _check(rc, "sl_comm_allgather")
        return recv

    def allreduce(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        """In place; fp32 / fp64 / int64, ``op`` "sum", "max" or "min"."""
        t = self._dev(t)
        with _on(self.device):
            rc = lib().sl_comm_allreduce(self._h, _ptr(t), t.numel(), _COMM_DTYPES[t.dtype], _COMM_OPS[op], _stream(t))
        _check(rc, "sl_comm_allreduce")
        return t

    def actmax_allgather_merge(self, states):
        """``states``: [(vals (C_l,k) bf16, ids (C_l,k) int64)] live device tensors, updated in place to the global top-k:
        pack -> ONE all-gather -> K4 per layer against the other ranks' blocks (``sl_actmax_allgather_merge``)."""
        if not states:
            return
        n, C, pv, pi, k = _state_tables(states)
        self._dev(states[0][0])
        need = lib().sl_actmax_allgather_merge_ws_bytes(n, C, k, self.world)
        ws = torch.empty(max(int(need), 16), dtype=torch.uint8, device=self.device)
        with _on(self.device):
            rc = lib().sl_actmax_allgather_merge(self._h, n, pv, pi, C, k, _ptr(ws), ws.numel(), _stream(ws))
        _check(rc, "sl_actmax_allgather_merge")
        ws.record_stream(torch.cuda.current_stream(self.device))


# ------------------------------------------------------------------------------------------------
# K5
# ------------------------------------------------------------------------------------------------
def gather_rows(emb: torch.Tensor, ids: torch.Tensor, check: bool = True) -> torch.Tensor:
    """``emb[ids]`` for emb (N,D) f32 on the device; negative ids wrap; out-of-range raises IndexError
    (``check=False`` skips that readback — and its host synchronisation — for indices known to be in range;
    out-of-range rows are then left unwritten)."""
    assert emb.is_cuda and emb.dtype == torch.float32 and emb.ndim == 2
    emb = emb.contiguous()
    ids_d = to_device(ids, emb.device).to(torch.int64).contiguous()
    N, D = emb.shape
    out = torch.empty(tuple(ids_d.shape) + (D,), dtype=torch.float32, device=emb.device)
    flag = torch.zeros(1, dtype=torch.int32, device=emb.device)
    with _on(emb.device):
        rc = lib().sl_gather_rows(_ptr(emb), N, D, _ptr(ids_d), ids_d.numel(), _ptr(out), _ptr(flag), _stream(emb))
    _check(rc, "sl_gather_rows")
    if check and int(flag.item()) != 0:
        raise IndexError(f"index out of range in embeds[sample_ids] (dataset size {N})")
    return out


def gather_rows_shard(emb_local: torch.Tensor, ids: torch.Tensor, row_offset: int, n_total: int) -> torch.Tensor:
    """Sharded ``emb[ids]``: rows this shard does not hold come back as zeros (sum over shards = full gather)."""
    assert emb_local.is_cuda and emb_local.dtype == torch.float32 and emb_local.ndim == 2
    emb_local = emb_local.contiguous()
    ids_d = to_device(ids, emb_local.device).to(torch.int64).contiguous()
    n_local, D = emb_local.shape
    out = torch.empty(tuple(ids_d.shape) + (D,), dtype=torch.float32, device=emb_local.device)
    flag = torch.zeros(1, dtype=torch.int32, device=emb_local.device)
    with _on(emb_local.device):
        rc = lib().sl_gather_rows_shard(
            _ptr(emb_local), n_local, D, _ptr(ids_d), ids_d.numel(), row_offset, n_total, _ptr(out), _ptr(flag),
            _stream(emb_local),
        )
    _check(rc, "sl_gather_rows_shard")
    if int(flag.item()) != 0:
        raise IndexError(f"index out of range in embeds[sample_ids] (dataset size {n_total})")
    return out


# ------------------------------------------------------------------------------------------------
# K6 / K7 / K8 / K10
# ------------------------------------------------------------------------------------------------
def _f32c(t: torch.Tensor, device=None) -> torch.Tensor:
    return to_device(t, device).to(torch.float32).contiguous()


def similarity(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    xd = _f32c(x)
    yd = _f32c(y, xd.device)
    if xd.ndim != 2 or yd.ndim != 2:
        raise ValueError("similarity_score expects 2-D tensors")
    xr, xc = xd.shape
    yr, yc = yd.shape
    if xd.shape == yd.shape:
        out = torch.empty((xr,), dtype=torch.float32, device=xd.device)
    elif xc == yr:
        out = torch.empty((xr, yc), dtype=torch.float32, device=xd.device)
    elif xc == yc:
        out = torch.empty((xr, yr), dtype=torch.float32, device=xd.device)
    else:
        raise ValueError("x and y must have the same shape")
    nbytes = int(lib().sl_similarity_ws_bytes(xr, xc, yr, yc))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xd.device)
    with _on(xd.device):
        rc = lib().sl_similarity(_ptr(xd), xr, xc, _ptr(yd), yr, yc, _ptr(out), _ptr(ws), nbytes, _stream(xd))
    _check(rc, "sl_similarity")
    return out


_policy_explicit = False  # the caller chose a cache policy (set_reduce_policy or the environment): the tuner keeps out


def _set_reduce_policy_raw(nt_min_bytes, tail_bytes):
    _check(lib().sl_set_reduce_policy(-1 if nt_min_bytes is None else int(nt_min_bytes),
                                      -1 if tail_bytes is None else int(tail_bytes)), "sl_set_reduce_policy")


def set_reduce_policy(nt_min_bytes: int | None = None, tail_bytes: int | None = None):
    """Cache policy of K1's row streams (include/semanticlens_amd.h ``sl_set_reduce_policy``): ``(0, 0)`` reads everything
    with the read-once policy (cold inputs), ``None`` restores the defaults (inputs just written by the previous kernel).
    An explicit choice also switches the per-layer :class:`ReducePolicyTuner` of the collect hooks off."""
    global _policy_explicit
    _policy_explicit = nt_min_bytes is not None or tail_bytes is not None
    _set_reduce_policy_raw(nt_min_bytes, tail_bytes)


def reduce_policy_is_explicit() -> bool:
    return _policy_explicit or "SL_NT_MIN_BYTES" in os.environ or "SL_REDUCE_TAIL_MB" in os.environ


class ReducePolicyTuner:
    """Which cache policy a hooked layer's reduce should read its input with — measured, per layer, on the first batches.

    The reduce kernels cannot know their producer.  The library default (inputs below 256 MiB and the last 240 MiB of larger
    ones with the default policy, the head read-once) is what a layer ending in an IN-PLACE activation wants (torchvision's
    Bottleneck: the whole output was just re-written, so the Infinity Cache holds its tail).  A layer whose output comes out of
    a residual ADD of two other tensors (transformer blocks, ConvNeXt) streamed three times its size through that cache: only
    ~80 MB of the output's tail are still there, and reading 240 MiB with the default policy thrashes (ConvNeXt-L's stage
    outputs: 0.625 of spec with the default, 0.693 with an 80 MiB tail; ViT-B/16's 155 MB block outputs 0.718 -> 0.742 with
    nt from 96 MiB: ``profiles/r04_reduce_policy_sweep.txt``).  So each layer tries the candidates on its first launches (one untimed +
    ``TRIALS`` timed each, HIP events read back only once they have completed: no synchronisation) and keeps the faster;
    results do not depend on the policy.  Off when the caller chose a policy (``set_reduce_policy`` / ``SL_NT_MIN_BYTES`` /
    ``SL_REDUCE_TAIL_MB``) or with ``SL_REDUCE_AUTOTUNE=0``; inputs below 96 MiB are never tuned."""

    # (nt_min_bytes, tail_bytes); None = the library default.  Round 5: behind a three-stream residual add the 617 MB ConvNeXt-L
    # stage reads 0.690 with an 80 MiB tail and 0.715 with 128 MiB (profiles/r05_k2_producer_lab.txt): a third candidate
    CANDIDATES = ((None, None), (96 << 20, 80 << 20), (96 << 20, 128 << 20))
    MIN_BYTES = 96 << 20
    TRIALS = 3

    def __init__(self):
        self.choice = None
        self._key = None
        self._calls = 0
        self._samples = [[] for _ in self.CANDIDATES]
        self._pending = []

    @staticmethod
    def enabled() -> bool:
        return os.environ.get("SL_REDUCE_AUTOTUNE", "1") != "0" and not reduce_policy_is_explicit()

    def run(self, launch, nbytes: int, batch: int):
        """``launch()`` enqueues the reduce on the current stream; called under the candidate / chosen policy."""
        if nbytes < self.MIN_BYTES or batch <= 0 or not self.enabled():
            return launch()
        key = nbytes // batch
        if key != self._key:  # another layer shape behind the same hook: start over
            self.__init__()
            self._key = key
        if self.choice is not None:
            if self.choice == 0:
                return launch()
            _set_reduce_policy_raw(*self.CANDIDATES[self.choice])
            try:
                return launch()
            finally:
                _set_reduce_policy_raw(None, None)
        idx = self._calls % len(self.CANDIDATES)
        timed = self._calls >= len(self.CANDIDATES)  # the first launch of each candidate also pays first-use costs
        self._calls += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _set_reduce_policy_raw(*self.CANDIDATES[idx])
        try:
            e0.record()
            out = launch()
            e1.record()
        finally:
            _set_reduce_policy_raw(None, None)
        if timed:
            self._pending.append((idx, e0, e1, nbytes))
        self._harvest()
        return out

    _registry: dict = {}

    @classmethod
    def for_site(cls, key) -> "ReducePolicyTuner":
        """The tuner of one call site (a hooked module): kept for the life of the process, so a second visualizer over the same
        model does not measure again."""
        t = cls._registry.get(key)
        if t is None:
            t = cls._registry[key] = cls()
        return t

    def _harvest(self):
        keep = []
        for idx, e0, e1, nbytes in self._pending:
            if e1.query():
                self._samples[idx].append(e0.elapsed_time(e1) / nbytes)
            else:
                keep.append((idx, e0, e1, nbytes))
        self._pending = keep
        if all(len(s_) >= self.TRIALS for s_ in self._samples):
            med = [sorted(s_)[len(s_) // 2] for s_ in self._samples]
            best = min(range(len(med)), key=med.__getitem__)
            self.choice = best if med[best] < 0.98 * med[0] else 0  # the default unless another is clearly faster
            self.medians_ns_per_mb = [m * 1e12 for m in med]  # ms per byte -> ns per MB
            self._pending = []


def set_gemm_mode(mode: str | None):
    """Arithmetic of the cosine GEMMs: "bf16x3" (default), "f32" (fp32-input MFMA) or None (environment)."""
    _check(lib().sl_set_gemm_mode({None: -1, "f32": 0, "bf16x3": 1}[mode]), "sl_set_gemm_mode")


def similarity_multi(x: torch.Tensor, ys: list[torch.Tensor]) -> list[torch.Tensor] | None:
    """``[similarity(x, y) for y in ys]`` with the query normalised/split once.  Returns None when a layer would
    take one of ``similarity_score``'s shape-quirk branches (the caller then goes layer by layer)."""
    if x.ndim != 2 or any(y.ndim != 2 or y.shape[1] != x.shape[1] for y in ys):
        return None
    Q, K = x.shape
    if any(y.shape[0] == Q or y.shape[0] == K for y in ys) or not ys:
        return None
    xd = _f32c(x)
    yds = [_f32c(y, xd.device) for y in ys]
    outs = [torch.empty((Q, y.shape[0]), dtype=torch.float32, device=xd.device) for y in yds]
    L = len(yds)
    cs = (_i64 * L)(*[y.shape[0] for y in yds])
    yp = (_vp * L)(*[y.data_ptr() if y.numel() else None for y in yds])
    op = (_vp * L)(*[o.data_ptr() if o.numel() else None for o in outs])
    nbytes = int(lib().sl_similarity_multi_ws_bytes(Q, K, cs, L))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xd.device)
    with _on(xd.device):
        rc = lib().sl_similarity_multi(_ptr(xd), Q, K, yp, cs, L, op, _ptr(ws), nbytes, _stream(xd))
    _check(rc, "sl_similarity_multi")
    return outs


def clarity(V: torch.Tensor) -> torch.Tensor:
    Vd = _f32c(V)
    lead = Vd.shape[:-2]
    n, D = Vd.shape[-2:]
    C = int(torch.tensor(lead).prod().item()) if len(lead) else 1
    out = torch.empty((C,), dtype=torch.float32, device=Vd.device)
    with _on(Vd.device):
        _check(lib().sl_clarity(_ptr(Vd), C, n, D, _ptr(out), _stream(Vd)), "sl_clarity")
    return out.reshape(lead)


def clarity_multi(Vs: list[torch.Tensor]) -> list[torch.Tensor] | None:
    """``[clarity(V) for V in Vs]`` as ONE launch when every ``V`` is ``(C_l, n, D)`` with the same ``(n, D)``; None otherwise
    (the caller then goes layer by layer)."""
    if not Vs or any(V.ndim != 3 for V in Vs) or len({tuple(V.shape[1:]) for V in Vs}) != 1:
        return None
    vds = [_f32c(Vs[0])]
    vds += [_f32c(V, vds[0].device) for V in Vs[1:]]
    n, D = vds[0].shape[1:]
    outs = [torch.empty((V.shape[0],), dtype=torch.float32, device=vds[0].device) for V in vds]
    L = len(vds)
    cs = (_i64 * L)(*[V.shape[0] for V in vds])
    vp = (_vp * L)(*[V.data_ptr() if V.numel() else None for V in vds])
    op = (_vp * L)(*[o.data_ptr() if o.numel() else None for o in outs])
    with _on(vds[0].device):
        _check(lib().sl_clarity_multi(vp, cs, L, n, D, op, _stream(vds[0])), "sl_clarity_multi")
    return outs


def redundancy(V: torch.Tensor) -> torch.Tensor:
    Vd = _f32c(V)
    lead = Vd.shape[:-2]
    C, D = Vd.shape[-2:]
    Bt = int(torch.tensor(lead).prod().item()) if len(lead) else 1
    out = torch.empty((Bt,), dtype=torch.float32, device=Vd.device)
    nbytes = int(lib().sl_redundancy_ws_bytes(Bt, C, D))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=Vd.device)
    with _on(Vd.device):
        _check(lib().sl_redundancy(_ptr(Vd), Bt, C, D, _ptr(out), _ptr(ws), nbytes, _stream(Vd)), "sl_redundancy")
    return out.reshape(lead)


def template_mean(E: torch.Tensor, E0: torch.Tensor, Q: int) -> torch.Tensor:
    Ed = _f32c(E)
    E0d = _f32c(E0, Ed.device)
    T, D = E0d.shape
    if Ed.shape != (Q * T, D):
        raise ValueError(f"templated embeddings have shape {tuple(Ed.shape)}, expected {(Q * T, D)}")
    out = torch.empty((Q, D), dtype=torch.float32, device=Ed.device)
    with _on(Ed.device):
        _check(lib().sl_template_mean(_ptr(Ed), _ptr(E0d), Q, T, D, _ptr(out), _stream(Ed)), "sl_template_mean")
    return out


# ------------------------------------------------------------------------------------------------
# measurement
# ------------------------------------------------------------------------------------------------
def prof_enable(on: bool = True):
    _check(lib().sl_prof_enable(1 if on else 0), "sl_prof_enable")


def prof_reset():
    _check(lib().sl_prof_reset(), "sl_prof_reset")


def prof_read(family: int):
    """(total_ms, launches, work) for one kernel family; synchronises the recorded events."""
    ms, n, w = ctypes.c_double(), _i64(), ctypes.c_double()
    _check(lib().sl_prof_read(family, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w)), "sl_prof_read")
    return ms.value, n.value, w.value


# ------------------------------------------------------------------------------------------------
# K9
# ------------------------------------------------------------------------------------------------
POLY2_MAX_N = 128  # sl_poly2means keeps a component's state in LDS; larger n / other k: sl_polykmeans


POLYK_MAX_N = 1024        # sl_polykmeans: samples per component (kMaxNGeneral in csrc/kmeans.hip)
POLYK_MAX_CLUSTERS = 16   # sl_polykmeans: n_clusters (kMaxClusters)
POLYK_MAX_COMPONENTS = 65535  # components per launch (grid.y of the Gram kernel)


def _poly_ws_budget() -> int:
    """Workspace bytes one K9 launch may take (default 8 GiB; ``SL_POLY_WS_GB``).  More components than fit run as
    consecutive launches over component chunks — every component is independent, so the scores do not change."""
    import os

    return int(float(os.environ.get("SL_POLY_WS_GB", "8")) * (1 << 30))


def poly2means(V: torch.Tensor, first_center, rand, replace_empty_clusters: bool = True, n_clusters: int = 2) -> torch.Tensor:
    """polysemanticity of V (C,n,D): k-means per component on the device; float64 (C,) result.

    Any number of components: calls are chunked by the Gram kernel's grid limit (65 535 components) and by the workspace
    budget.  ``n_samples > 1024`` with ``n_clusters != 2`` (or > 128 samples), and ``n_clusters > 16``, raise
    ``ValueError`` — limits the reference (scikit-learn on the host, scores.py:131-185) does not have."""
    import numpy as np

    Vd = _f32c(V)
    if Vd.ndim != 3:
        raise ValueError("polysemanticity_score expects a (n_components, n_samples, n_features) tensor")
    C, n, D = Vd.shape
    first_center = np.ascontiguousarray(first_center, dtype=np.int32)
    rand = np.ascontiguousarray(rand, dtype=np.float64)
    n_init = int(first_center.shape[0])
    out = torch.empty((C,), dtype=torch.float64, device=Vd.device)
    mincnt = torch.empty((C,), dtype=torch.int32, device=Vd.device)
    general = n_clusters != 2 or n > POLY2_MAX_N
    if general:
        if n_clusters > POLYK_MAX_CLUSTERS:
            raise ValueError(f"n_clusters={n_clusters} exceeds the device kernel's maximum of {POLYK_MAX_CLUSTERS}")
        if n > POLYK_MAX_N:
            raise ValueError(f"n_samples={n} exceeds the device kernel's maximum of {POLYK_MAX_N} samples per component")
    if C == 0:
        return out
    per_comp = int(lib().sl_polykmeans_ws_bytes(1, n, D, n_clusters, n_init)) if general else int(lib().sl_poly2means_ws_bytes(1, n, D))
    chunk = max(1, min(C, POLYK_MAX_COMPONENTS, _poly_ws_budget() // max(per_comp, 1)))
    ws = None
    with _on(Vd.device):
        for c0 in range(0, C, chunk):
            cc = min(chunk, C - c0)
            Vc, oc, mc = Vd[c0:c0 + cc], out[c0:c0 + cc], mincnt[c0:c0 + cc]
            if general:
                nbytes = int(lib().sl_polykmeans_ws_bytes(cc, n, D, n_clusters, n_init))
                if ws is None or ws.numel() < nbytes:
                    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=Vd.device)
                rc = lib().sl_polykmeans(
                    _ptr(Vc), cc, n, D, int(n_clusters), first_center.ctypes.data_as(_vp), n_init, rand.ctypes.data_as(_vp),
                    1 if replace_empty_clusters else 0, _ptr(oc), _ptr(mc), _ptr(ws), nbytes, _stream(Vd),
                )
                torch.cuda.current_stream(Vd.device).synchronize()  # the draws were copied from host arrays owned by this call
                _check(rc, "sl_polykmeans")
            else:
                nbytes = int(lib().sl_poly2means_ws_bytes(cc, n, D))
                if ws is None or ws.numel() < nbytes:
                    ws = torch.empty(nbytes, dtype=torch.uint8, device=Vd.device)
                rc = lib().sl_poly2means(
                    _ptr(Vc), cc, n, D, first_center.ctypes.data_as(_vp), n_init, rand.ctypes.data_as(_vp),
                    1 if replace_empty_clusters else 0, _ptr(oc), _ptr(mc), _ptr(ws), nbytes, _stream(Vd),
                )
                if rc == -3:
                    raise NotImplementedError(lib().sl_last_error().decode())
                _check(rc, "sl_poly2means")
    return out


# ------------------------------------------------------------------------------------------------
# K11: native transformer-tower primitives (all on torch's current stream, fp32, contiguous)
# ------------------------------------------------------------------------------------------------
def linear(x, w, bias=None, act=SL_ACT_NONE, residual=None, out=None, scatter=None, rowadd=None):
    """out = act(x @ w.T + bias) (+ residual).  x (M,K), w (N,K).  ``scatter=(rows_per_group, group_stride,
    row_offset)`` writes row r to (r // rpg) * group_stride + row_offset + r % rpg of ``out`` (+ ``rowadd`` rows)."""
    M, K = x.shape
    Nn = w.shape[0]
    _need_f32("linear", x, w, bias, residual, out, rowadd)
    if out is None:
        out = torch.empty((M, Nn), dtype=torch.float32, device=x.device)
    rpg, gs, ro = scatter if scatter is not None else (0, 0, 0)
    with _on(x.device):
        rc = lib().sl_linear(_ptr(x), M, K, _ptr(w), Nn, _ptr(bias), act, _ptr(residual), _ptr(out), out.stride(0) if out.ndim == 2 else Nn,
                             rpg, gs, ro, _ptr(rowadd), _stream(x))
    _check(rc, "sl_linear")
    return out


class Split:
    """An fp32 matrix carried as a split matrix (hi = bf16(v), lo = bf16(v - hi)): operand of the bf16x3 GEMM.

    One buffer of ``rows x 2 Kp`` bf16 values, ``Kp`` = ``cols`` rounded up to 32 and zero padded; each 32-wide k-tile of a
    row is one 128-byte line ``[hi(32) | lo(32)]`` (``include/semanticlens_amd.h``).  ``hi`` / ``lo`` materialise the two
    halves as ``(rows, cols)`` bf16 tensors (tests, debugging)."""

    def __init__(self, rows: int, cols: int, device):
        self.shape = (rows, cols)
        self.kp = (cols + 31) // 32 * 32
        # producers write columns < cols only; the padding (if any) must read as zero in the GEMM
        alloc = torch.empty if self.kp == cols else torch.zeros
        self.buf = alloc((rows, 2 * self.kp), dtype=torch.bfloat16, device=device)
        assert self.buf.numel() == int(lib().sl_split_elems(rows, cols))

    def _half(self, which: int) -> torch.Tensor:
        rows, cols = self.shape
        return self.buf.view(rows, self.kp // 32, 2, 32)[:, :, which, :].reshape(rows, self.kp)[:, :cols]

    @property
    def hi(self) -> torch.Tensor:
        return self._half(0)

    @property
    def lo(self) -> torch.Tensor:
        return self._half(1)

    @classmethod
    def of(cls, x: torch.Tensor, row_scale: torch.Tensor | None = None) -> "Split":
        _need_f32("Split.of", x, row_scale)
        x = x.contiguous()
        out = cls(x.shape[0], x.shape[1], x.device)
        with _on(x.device):
            rc = lib().sl_split_bf16(_ptr(x), _ptr(row_scale), x.shape[0], x.shape[1], _ptr(out.buf), _stream(x))
        _check(rc, "sl_split_bf16")
        return out


def _split_ptr(sp):
    return _ptr(sp.buf) if sp is not None else _vp(None)


def linear3(x: Split, w: Split, bias=None, act=SL_ACT_NONE, residual=None, out=None, out_split: Split | None = None,
            scatter=None, rowadd=None):
    """bf16x3 version of :func:`linear`: x, w are :class:`Split` operands; the result is fp32 ``out`` (optionally
    + residual, or scattered rows) or, with ``out_split``, split bf16 ready to feed the next GEMM."""
    M, K = x.shape
    Nn = w.shape[0]
    dev = x.buf.device
    _need_f32("linear3", bias, residual, out, rowadd)
    if w.shape[1] != K:
        raise ValueError(f"linear3: x has {K} columns, w has {w.shape[1]}")
    if out is None and out_split is None:
        out = torch.empty((M, Nn), dtype=torch.float32, device=dev)
    ldo = out.stride(0) if out is not None else Nn
    rpg, gs, ro = scatter if scatter is not None else (0, 0, 0)
    with _on(dev):
        rc = lib().sl_linear_bf16x3(_ptr(x.buf), M, K, _ptr(w.buf), Nn, _ptr(bias), act, _ptr(residual), _ptr(out),
                                    _split_ptr(out_split), ldo, rpg, gs, ro, _ptr(rowadd), _stream(x.buf))
    _check(rc, "sl_linear_bf16x3")
    return out if out is not None else out_split


def layernorm(x, gamma, beta, eps, out=None, rows=None, x_row_stride=None, out_split: Split | None = None):
    _need_f32("layernorm", x, gamma, beta, out)
    cols = x.shape[-1]
    rows = rows if rows is not None else x.numel() // cols
    xs = x_row_stride if x_row_stride is not None else cols
    if out is None and out_split is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    with _on(x.device):
        rc = lib().sl_layernorm(_ptr(x), rows, cols, xs, _ptr(gamma), _ptr(beta), float(eps), _ptr(out), _split_ptr(out_split),
                                cols, _stream(x))
    _check(rc, "sl_layernorm")
    return out if out is not None else out_split


def attention(qkv, B, T, H, head_dim, causal, out=None, out_split: Split | None = None, bf16x3: bool = False):
    """softmax(q k^T / sqrt(d)) v per (batch, head).  ``bf16x3``: both products in the split-bf16 x3 arithmetic of
    :func:`linear3` (``sl_attention_bf16x3``) instead of fp32-input MFMAs."""
    _need_f32("attention", qkv, out)
    if out is None and out_split is None:
        out = torch.empty((B * T, H * head_dim), dtype=torch.float32, device=qkv.device)
    fn = lib().sl_attention_bf16x3 if bf16x3 else lib().sl_attention
    with _on(qkv.device):
        rc = fn(_ptr(qkv), B, T, H, head_dim, 1 if causal else 0, _ptr(out), _split_ptr(out_split), _stream(qkv))
    _check(rc, "sl_attention_bf16x3" if bf16x3 else "sl_attention")
    return out if out is not None else out_split


def attention_pool(q, kv, B, T, H, head_dim, out=None):
    """One query per head against the keys / values of each image: ``q`` (H*head_dim) projected probe, ``kv`` (B*T, 2*W)
    rows ``[k | v]`` -> ``(B, W)``."""
    W = H * head_dim
    _need_f32("attention_pool", q, kv, out)
    if out is None:
        out = torch.empty((B, W), dtype=torch.float32, device=kv.device)
    with _on(kv.device):
        rc = lib().sl_attention_pool(_ptr(q), _ptr(kv), kv.stride(0), W, B, T, H, head_dim, _ptr(out), _stream(kv))
    _check(rc, "sl_attention_pool")
    return out


def attention_pool_q(q, kv, B, T, H, head_dim, out=None):
    """``attention_pool`` with one query per image: ``q`` (B, H*head_dim) -> ``(B, W)`` (CLIP-ResNet attention pool)."""
    W = H * head_dim
    _need_f32("attention_pool_q", q, kv, out)
    if out is None:
        out = torch.empty((B, W), dtype=torch.float32, device=kv.device)
    with _on(kv.device):
        rc = lib().sl_attention_pool_q(_ptr(q), q.stride(0), _ptr(kv), kv.stride(0), W, B, T, H, head_dim, _ptr(out), _stream(kv))
    _check(rc, "sl_attention_pool_q")
    return out


def tokens_from_map(fmap, pos, out=None):
    """(B, C, H, W) trunk output + (HW + 1, C) positions -> (B, HW + 1, C) token rows, row 0 = mean token (``sl_tokens_from_map``)."""
    B, C = fmap.shape[:2]
    S = fmap[0, 0].numel()
    fmap = fmap.contiguous()
    _need_f32("tokens_from_map", fmap, pos, out)
    if out is None:
        out = torch.empty((B, S + 1, C), dtype=torch.float32, device=fmap.device)
    with _on(fmap.device):
        rc = lib().sl_tokens_from_map(_ptr(fmap), B, C, S, _ptr(pos), _ptr(out), _stream(fmap))
    _check(rc, "sl_tokens_from_map")
    return out


def patchify(img, P, out=None, out_split: Split | None = None):
    B, C, Hi, Wi = img.shape
    _need_f32("patchify", img, out)
    if out is None and out_split is None:
        out = torch.empty((B * (Hi // P) * (Wi // P), C * P * P), dtype=torch.float32, device=img.device)
    with _on(img.device):
        rc = lib().sl_patchify(_ptr(img), B, C, Hi, Wi, P, _ptr(out), _split_ptr(out_split), _stream(img))
    _check(rc, "sl_patchify")
    return out if out is not None else out_split


def broadcast_row(v, add, G, group_stride_elems, out):
    with _on(out.device):
        rc = lib().sl_broadcast_row(_ptr(v), _ptr(add), G, group_stride_elems, v.numel(), _ptr(out), _stream(out))
    _check(rc, "sl_broadcast_row")


def embed_tokens(table, ids, pos, out=None):
    B, T = ids.shape
    vocab, W = table.shape
    if out is None:
        out = torch.empty((B * T, W), dtype=torch.float32, device=table.device)
    with _on(table.device):
        rc = lib().sl_embed_tokens(_ptr(table), vocab, _ptr(ids), B, T, W, _ptr(pos), _ptr(out), _stream(table))
    _check(rc, "sl_embed_tokens")
    return out


# ------------------------------------------------------------------------------------------------
# K12: image preprocessing (foundation_models/clip.py:157-163 of the reference runs it per sample on the host)
# ------------------------------------------------------------------------------------------------
def preprocess_plan(hw, size: int, resize_mode: str = "shortest", interp: str = "bicubic", pixel_offsets=None):
    """Host-side plan for a ragged batch: ``hw`` (B, 2) heights/widths -> (plan (B, 16) int64 CPU tensor, info dict).
    Pure host arithmetic inside the library (no device needed)."""
    hw_t = torch.as_tensor(hw, dtype=torch.int32).reshape(-1, 2).contiguous()
    B = hw_t.shape[0]
    plan = torch.zeros((B, SL_PP_PLAN_STRIDE), dtype=torch.int64)
    info = (ctypes.c_int64 * 4)()
    off = None if pixel_offsets is None else torch.as_tensor(pixel_offsets, dtype=torch.int64).contiguous()
    if resize_mode not in PP_RESIZE_MODES or interp not in PP_INTERP:
        raise ValueError(f"unknown resize_mode/interpolation {resize_mode!r}/{interp!r}")
    _check(lib().sl_preprocess_plan(_vp(hw_t.data_ptr()) if B else _vp(None), _vp(off.data_ptr()) if off is not None and B else _vp(None),
                                    B, int(size), PP_RESIZE_MODES[resize_mode], PP_INTERP[interp],
                                    _vp(plan.data_ptr()) if B else _vp(None), ctypes.cast(info, _vp)), "sl_preprocess_plan")
    return plan, {"ws_bytes": int(info[0]), "coef_bytes": int(info[1]), "max_h": int(info[2]), "pixel_bytes": int(info[3])}


def preprocess(pixels: torch.Tensor, plan: torch.Tensor, info: dict, size: int, mean, std, interp: str = "bicubic",
               want_u8: bool = False, want_f32: bool = True):
    """Packed raw RGB bytes on the device + uploaded plan -> (B, 3, size, size) fp32 (and/or (B, size, size, 3) uint8)."""
    if not pixels.is_cuda or pixels.dtype != torch.uint8:
        raise TypeError("pixels must be a uint8 tensor on a HIP device")
    dev = pixels.device
    B = plan.shape[0]
    plan_d = plan.to(dev, non_blocking=True) if not plan.is_cuda else plan
    out = torch.empty((B, 3, size, size), dtype=torch.float32, device=dev) if want_f32 else None
    out_u8 = torch.empty((B, size, size, 3), dtype=torch.uint8, device=dev) if want_u8 else None
    ws = torch.empty((max(info["ws_bytes"], 16),), dtype=torch.uint8, device=dev)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std])
    with _on(dev):
        _check(lib().sl_preprocess(_ptr(pixels), _ptr(plan_d), B, int(size), PP_INTERP[interp], info["max_h"], info["coef_bytes"],
                                   ctypes.cast(m, _vp), ctypes.cast(sd, _vp), _ptr(out), _ptr(out_u8), _ptr(ws),
                                   info["ws_bytes"], _stream(pixels)), "sl_preprocess")
    return out, out_u8
