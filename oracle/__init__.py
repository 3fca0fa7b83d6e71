"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

numpy-facing wrapper around ``oracle/sl_oracle.cpp``, the CPU restatement of the
reference's concept-DB hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package.  The product
package ``semanticlens_amd`` never imports it.

Parity status: pinned against fixtures generated from the unmodified reference
(``tests/golden/make_golden.py``) — see ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libsl_oracle.so"
_lib = None

AGG_CONV = {"max": 0, "mean": 1, "sum": 2}
AGG_TOK = {"mean": 0, "absmean": 1, "max": 2, "absmax": 3, "token": 4}
MODE_ATEN = 0
MODE_TOTAL = 1


def build(force: bool = False) -> Path:
    """Compile the oracle shared library with g++ (``make -C oracle``)."""
    newest = max((_HERE / f).stat().st_mtime for f in ("sl_oracle.cpp", "sl_oracle_preprocess.cpp"))
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < newest:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
    return _lib


def set_threads(n: int):
    """Number of OpenMP threads the oracle uses (for the cpu_baseline timing)."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


_I = ctypes.c_int64


def f32_to_bf16(x) -> np.ndarray:
    x = _f32(x)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().orc_f32_to_bf16(_p(x), _p(out), _I(x.size))
    return out


def bf16_to_f32(h) -> np.ndarray:
    h = np.ascontiguousarray(h, dtype=np.uint16)
    return (h.astype(np.uint32) << 16).view(np.float32)


def agg_conv(x, agg: str) -> np.ndarray:
    """aggregators.aggregate_conv_{max,mean}: (B,C,H,W) -> (B,C)."""
    x = _f32(x)
    B, C = x.shape[:2]
    S = int(np.prod(x.shape[2:]))
    out = np.empty((B, C), dtype=np.float32)
    lib().orc_agg_conv(_p(x), _I(B), _I(C), _I(S), ctypes.c_int(AGG_CONV[agg]), _p(out))
    return out


def abs_norm_rows(x, eps: float = 1e-10) -> np.ndarray:
    """zennit-crp ``abs_norm``: rows of a (B, C) matrix divided by ``|row|.sum() + eps`` (sl_oracle.cpp::orc_abs_norm_rows)."""
    x = _f32(x).copy()
    lib().orc_abs_norm_rows(_p(x), _I(x.shape[0]), _I(x.shape[1]), ctypes.c_float(eps))
    return x


def agg_tokens(x, agg: str, pos: int = 0) -> np.ndarray:
    """aggregators.aggregate_transformer_*: (B,T,F) -> (B,F)."""
    x = _f32(x)
    B, T, F = x.shape
    out = np.empty((B, F), dtype=np.float32)
    lib().orc_agg_tokens(_p(x), _I(B), _I(T), _I(F), ctypes.c_int(AGG_TOK[agg]), _I(pos), _p(out))
    return out


class ActMaxOracle:
    """ActMax (activation_caching.py:64-141) with bf16 values kept as uint16 bit patterns."""

    def __init__(self, n_collect: int, n_latents: int, mode: int = MODE_ATEN, init_value: float | None = None):
        self.k, self.C, self.mode = int(n_collect), int(n_latents), mode
        self.vals = np.empty((self.C, self.k), dtype=np.uint16)
        self.ids = np.empty((self.C, self.k), dtype=np.int64)
        lib().orc_actmax_init(_p(self.vals), _p(self.ids), _I(self.C), _I(self.k))
        if init_value is not None:  # the build's relevance visualizer starts signed quantities at -inf (reference: -0.0)
            self.vals[:] = f32_to_bf16(np.full((1,), init_value, dtype=np.float32))[0]

    def update(self, acts, sample_ids):
        acts = _f32(acts)
        sample_ids = _i64(sample_ids)
        B, C = acts.shape
        assert C == self.C and sample_ids.shape == (B,)
        lib().orc_actmax_update(
            _p(self.vals), _p(self.ids), _I(self.C), _I(self.k), _p(acts), _p(sample_ids), _I(B), ctypes.c_int(self.mode)
        )

    def merge_states(self, other_vals, other_ids):
        ov = np.ascontiguousarray(other_vals, dtype=np.uint16)
        oi = _i64(other_ids)
        R = ov.shape[0]
        assert ov.shape == (R, self.C, self.k) and oi.shape == ov.shape
        lib().orc_actmax_merge_states(_p(self.vals), _p(self.ids), _I(self.C), _I(self.k), _p(ov), _p(oi), _I(R))

    @property
    def activations_f32(self):
        return bf16_to_f32(self.vals)


def gather_rows(emb, ids) -> np.ndarray:
    emb = _f32(emb)
    ids = _i64(ids)
    N, D = emb.shape
    out = np.empty(ids.shape + (D,), dtype=np.float32)
    rc = lib().orc_gather_rows(_p(emb), _I(N), _I(D), _p(ids), _I(ids.size), _p(out))
    if rc != 0:
        raise IndexError("index out of range")
    return out


def similarity(x, y) -> np.ndarray:
    x, y = _f32(x), _f32(y)
    if x.shape == y.shape:
        out = np.empty(x.shape[:1], dtype=np.float32)
    elif x.shape[1] == y.shape[0]:
        out = np.empty((x.shape[0], y.shape[1]), dtype=np.float32)
    elif x.shape[1] == y.shape[1]:
        out = np.empty((x.shape[0], y.shape[0]), dtype=np.float32)
    else:
        raise ValueError("x and y must have the same shape")
    rc = lib().orc_similarity(_p(x), _I(x.shape[0]), _I(x.shape[1]), _p(y), _I(y.shape[0]), _I(y.shape[1]), _p(out))
    assert rc >= 0
    return out


def similarity_torch(x, y):
    """`similarity_score` as the reference runs it on the host — `F.normalize` + `matmul` / `cosine_similarity` on torch-CPU
    tensors (scores.py:119-128) — for the cpu_baseline TIMING of `text_probing` (torch's sgemm, not this file's fp64 loops,
    is what the reference's CPU path costs).  Pinned to `similarity` / the golden vectors in tests/test_oracle_golden.py."""
    import torch
    import torch.nn.functional as F

    x, y = torch.as_tensor(x), torch.as_tensor(y)
    if x.shape == y.shape:
        return F.cosine_similarity(x, y, dim=-1)
    xn, yn = F.normalize(x, dim=-1), F.normalize(y, dim=-1)
    if x.shape[1] == y.shape[0]:
        return xn @ yn
    if x.shape[1] == y.shape[1]:
        return xn @ yn.T
    raise ValueError("x and y must have the same shape")


def clarity_torch(V):
    """`clarity_score` as the reference computes it on torch-CPU tensors (scores.py:45-46), for cpu_baseline timing."""
    import torch
    import torch.nn.functional as F

    V = torch.as_tensor(V)
    n = V.shape[-2]
    m = F.normalize(V, dim=-1).mean(-2)
    return ((m**2).sum(-1) - 1 / n) / (n - 1) * n


def clarity(V) -> np.ndarray:
    V = _f32(V)
    C, n, D = V.shape
    out = np.empty((C,), dtype=np.float32)
    lib().orc_clarity(_p(V), _I(C), _I(n), _I(D), _p(out))
    return out


def redundancy(V) -> np.ndarray:
    V = _f32(V)
    lead = V.shape[:-2]
    C, D = V.shape[-2:]
    Bt = int(np.prod(lead)) if lead else 1
    out = np.empty((Bt,), dtype=np.float32)
    lib().orc_redundancy(_p(V), _I(Bt), _I(C), _I(D), _p(out))
    return out.reshape(lead)


def template_mean(E, E0, Q: int) -> np.ndarray:
    E, E0 = _f32(E), _f32(E0)
    T, D = E0.shape
    assert E.shape == (Q * T, D)
    out = np.empty((Q, D), dtype=np.float32)
    lib().orc_template_mean(_p(E), _p(E0), _I(Q), _I(T), _I(D), _p(out))
    return out


def polysemanticity(V, random_state: int = 123, n_clusters: int = 2, return_fallback: bool = False):
    """polysemanticity_score — scores.py:131-185.

    The clustering itself lives in a third-party dependency of the reference
    (scikit-learn ``KMeans(n_clusters=2, n_init=10, random_state=123)``,
    scores.py:167; 1.7.2 in this image, 1.6.1/1.7.1 in the reference's uv.lock).
    The oracle calls that same dependency and restates everything around it.
    """
    from sklearn.cluster import KMeans

    V = _f32(V)
    C, n, D = V.shape
    # The reference hands sklearn a torch tensor (scores.py:167); torch dtypes have no `.kind`, so
    # sklearn's check_array falls back to its first accepted dtype and clusters in FLOAT64.
    fits = [KMeans(n_clusters=n_clusters, n_init=10, random_state=random_state).fit(e.astype(np.float64)) for e in V]
    centers = np.stack([f.cluster_centers_ for f in fits], 0)  # float64, like the reference's c_centers
    cn = centers / np.maximum(np.linalg.norm(centers, axis=-1, keepdims=True), 1e-12)
    poly = 1.0 - ((cn.mean(-2) ** 2).sum(-1) - 1.0 / n_clusters) / (n_clusters - 1) * n_clusters  # clarity_score in f64
    counts = []
    for f in fits:
        cnt = np.unique(f.labels_, return_counts=True)[1]
        counts.append(cnt if len(cnt) == n_clusters else np.zeros(n_clusters))
    bad = np.stack(counts, 0).min(-1) < 2  # scores.py:176-178
    if bad.any():
        v_not = V[bad]
        ns = min(10, n)
        acc = np.zeros(v_not.shape[0], dtype=np.float32)
        for i in range(ns):  # scores.py:182-183
            acc += clarity(np.stack([v_not.mean(1), v_not[:, i]], axis=1))
        poly[bad] = 1.0 - acc.astype(np.float64) / ns
    if return_fallback:  # which rows took the fp32 fallback branch (scores.py:178-184) — tests classify by it
        return poly, bad
    return poly


RESIZE_MODE = {"shortest": 0, "squash": 1}
INTERP = {"bicubic": 0, "bilinear": 1}


def resized_size(h: int, w: int, size: int, resize_mode: str = "shortest") -> tuple[int, int]:
    """torchvision ``Resize(size)`` output (h, w) — sl_oracle_preprocess.cpp::orc_resized_size."""
    oh, ow = ctypes.c_int(), ctypes.c_int()
    lib().orc_resized_size(int(h), int(w), int(size), RESIZE_MODE[resize_mode], ctypes.byref(oh), ctypes.byref(ow))
    return oh.value, ow.value


def center_crop_offset(size: int, crop: int) -> int:
    return int(lib().orc_center_crop_offset(int(size), int(crop)))


def preprocess(img, size: int, mean, std, resize_mode: str = "shortest", interp: str = "bicubic"):
    """One (h, w, 3) uint8 image -> ((size, size, 3) uint8 resized+cropped, (3, size, size) fp32 normalised):
    open_clip's inference transform (``clip.py:157-163`` calls it per sample) restated on the CPU."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3
    u8 = np.empty((size, size, 3), np.uint8)
    f = np.empty((3, size, size), np.float32)
    m, s = _f32(mean), _f32(std)
    rc = lib().orc_preprocess(_p(img), int(img.shape[0]), int(img.shape[1]), int(size), RESIZE_MODE[resize_mode],
                              INTERP[interp], _p(m), _p(s), _p(u8), _p(f))
    if rc != 0:
        raise ValueError("resized image smaller than the crop")
    return u8, f
