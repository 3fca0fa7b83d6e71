"""Naming helper used by the cache paths (reference: utils/helper.py:11-18)."""
from __future__ import annotations

import hashlib


def get_fallback_name(obj) -> str:
    """``<ClassName>-<int(sha256(str(obj)))>`` — identical to the reference so cache directories interchange."""
    return obj.__class__.__name__ + "-" + str(int(hashlib.sha256(str(obj).encode()).hexdigest(), 16))
