"""Concept-quality scores (reference: semanticlens/scores.py:18-185) on HIP kernels.

Same names, arguments and shape conventions as the reference — including the shape-dependent
branches of ``similarity_score`` (scores.py:119-128).  Inputs may live on the host or the
device; results come back on the input's device, as with the reference.  There is no CPU
implementation behind these functions: without a HIP device they raise.
"""
from __future__ import annotations

import logging

import numpy as np
import torch

from semanticlens_amd import _native as N

logger = logging.getLogger(__name__)


@torch.inference_mode()
def clarity_score(V):
    """Mean off-diagonal cosine among each component's samples.  V (..., n_samples, D) -> (...).

    Reference: scores.py:18-47 — ``((mean_j normalize(v_j))**2).sum(-1) - 1/n) / (n-1) * n``.  Kernel K7.
    """
    return N.clarity(V).to(V.device)


@torch.inference_mode()
def redundancy_score(cones):
    """Mean over components of the max cosine to any *other* component.  (..., C, D) -> (...).

    Reference: scores.py:50-81 (diagonal removed by ``- 2 * eye``).  Kernel K8 = K6's MFMA GEMM + row max.
    """
    return N.redundancy(cones).to(cones.device)


@torch.inference_mode()
def similarity_score(x, y):
    """Cosine similarity with the reference's shape branches (scores.py:84-128):

    * ``x.shape == y.shape``        -> row-wise cosine, shape ``(n,)``
    * ``x.shape[1] == y.shape[0]``  -> ``normalize(x) @ normalize(y)`` (no transpose)
    * ``x.shape[1] == y.shape[1]``  -> ``normalize(x) @ normalize(y).T``  (the probing GEMM, K6)
    * otherwise ``ValueError("x and y must have the same shape")``
    """
    return N.similarity(x, y).to(x.device)


def kmeans_draws(n_samples: int, n_init: int, random_state: int, n_clusters: int = 2):
    """The random draws scikit-learn's ``KMeans(n_clusters, n_init, random_state)`` consumes, per init.

    ``KMeans.fit`` seeds one ``RandomState(random_state)`` and, for each of the ``n_init`` k-means++ seedings, draws
    ``choice(n_samples, p=uniform)`` for the first centre and, for every further centre,
    ``uniform(size=2 + int(log(n_clusters)))`` for its candidates (sklearn/cluster/_kmeans.py ``_kmeans_plusplus``).
    None of it depends on the data, so the host produces the draws and the device kernel consumes them.
    Returns ``first (n_init,) int32`` and ``rand (n_init, n_clusters - 1, trials) float64``.
    """
    rs = np.random.RandomState(random_state)
    weights = np.ones(n_samples, dtype=np.float64)  # KMeans' sample_weight takes X's dtype: float64 (see oracle)
    p = weights / weights.sum()
    trials = 2 + int(np.log(n_clusters))
    first = np.empty(n_init, dtype=np.int32)
    rand = np.empty((n_init, max(n_clusters - 1, 1), trials), dtype=np.float64)
    for i in range(n_init):
        first[i] = rs.choice(n_samples, p=p)
        for c in range(n_clusters - 1):
            rand[i, c] = rs.uniform(size=trials)
    return first, rand


@torch.inference_mode()
def polysemanticity_score(V, replace_empty_clusters=True, random_state=123, n_clusters=2):
    """``1 - clarity_score(centres)`` of a k-means clustering of each component's samples (``n_clusters`` = 2 by default:
    ``1 - cos(centre_1, centre_2)``).

    Reference: scores.py:131-185 (scikit-learn ``KMeans(n_clusters, n_init=10, random_state=123)`` per component in a
    Python loop).  Kernel K9 restates that procedure per component on the device for ``n_clusters`` up to 16 and up to
    1024 samples per component; rows whose smallest cluster has fewer than 2 samples use the reference's fallback.
    ``n_samples < n_clusters`` raises ``ValueError`` as scikit-learn does.
    """
    first, rand = kmeans_draws(V.shape[-2], 10, random_state, n_clusters)
    return N.poly2means(V, first, rand, replace_empty_clusters, n_clusters=n_clusters).to(V.device)
