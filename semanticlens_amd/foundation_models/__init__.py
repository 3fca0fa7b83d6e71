"""Foundation-model plugins (reference: foundation_models/__init__.py:12-14) plus the native execution paths."""
from semanticlens_amd.foundation_models.base import AbstractVLM
from semanticlens_amd.foundation_models.clip import ClipMobile, OpenClip, SigLipV2
from semanticlens_amd.foundation_models.native_clip import NativeClip, NativeSigLip, NativeTextClip
from semanticlens_amd.foundation_models.preprocess import DevicePreprocess

__all__ = ["AbstractVLM", "OpenClip", "ClipMobile", "SigLipV2", "DevicePreprocess", "NativeClip", "NativeSigLip", "NativeTextClip"]
