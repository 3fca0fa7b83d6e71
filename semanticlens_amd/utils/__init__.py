"""Utilities on the concept-DB hot path (the reference's plotting/logging helpers are out of scope)."""
from semanticlens_amd.utils.helper import get_fallback_name

__all__ = ["get_fallback_name"]
