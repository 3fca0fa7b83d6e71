"""Utilities on the concept-DB hot path (reference: utils/__init__.py:16-24).  ``to_transforms_compose`` (a torchvision
preset converter) and ``setup_colored_logging`` are outside the path and not provided."""
from semanticlens_amd.utils.helper import get_denormalization_transform, get_fallback_name

__all__ = ["get_fallback_name", "get_denormalization_transform"]
