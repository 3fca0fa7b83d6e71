"""Component visualizers (reference: component_visualization/__init__.py:16-22).

``RelevanceComponentVisualizer`` of the reference wraps zennit-crp (not installed here) and is declared broken upstream
(relevance_based.py:27); the class of the same name here restates its procedure on this package's kernels (parity
unpinned, see its module docstring).
"""
from semanticlens_amd.component_visualization.activation_based import ActivationComponentVisualizer
from semanticlens_amd.component_visualization.relevance_based import RelevanceComponentVisualizer

__all__ = ["ActivationComponentVisualizer", "RelevanceComponentVisualizer"]
