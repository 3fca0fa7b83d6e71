"""Utilities (reference: utils/__init__.py:16-24).  ``to_transforms_compose`` (a torchvision preset converter) is not
provided: torchvision is not a dependency."""
from semanticlens_amd.utils.helper import get_denormalization_transform, get_fallback_name
from semanticlens_amd.utils.log_setup import setup_colored_logging

__all__ = ["get_fallback_name", "get_denormalization_transform", "setup_colored_logging"]
