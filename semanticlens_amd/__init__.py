"""semanticlens_amd — the SemanticLens concept-database hot path on AMD MI355X (gfx950).

Drop-in for ``semanticlens`` (jim-berend/semanticlens v0.2.1) on that path: the same
``Lens`` / ``ActivationComponentVisualizer`` / ``foundation_models`` / ``scores`` API
(reference ``semanticlens/__init__.py:35-47``) over hand-written HIP kernels reached through
the C ABI in ``include/semanticlens_amd.h``.  No CPU fallback: without the built library and a
HIP device, compute calls raise.
"""
from __future__ import annotations

from semanticlens_amd import foundation_models, scores, utils
from semanticlens_amd.lens import Lens
from semanticlens_amd.scores import clarity_score, polysemanticity_score, redundancy_score

__version__ = "0.1.0"

__all__ = [
    "foundation_models",
    "scores",
    "utils",
    "Lens",
    "clarity_score",
    "polysemanticity_score",
    "redundancy_score",
]
