"""Component visualizers (reference: component_visualization/__init__.py:16-22).

``RelevanceComponentVisualizer`` of the reference wraps zennit-crp and is declared broken
upstream (relevance_based.py:27; abstract members missing, SURVEY.md finding 5); it is out of
scope for the concept-DB hot path and not provided.
"""
from semanticlens_amd.component_visualization.activation_based import ActivationComponentVisualizer

__all__ = ["ActivationComponentVisualizer"]
