"""OpenCLIP-family foundation models behind ``AbstractVLM``.

API mirror of ``semanticlens/foundation_models/clip.py:27-247`` (``OpenClip(url, device)``, ``SigLipV2(device)``,
``ClipMobile(version, device)``; ``encode_image`` / ``encode_text`` return UN-normalised features, ``preprocess``
always returns a batch, ``tokenize`` pads to the model's context length).  The encoder arithmetic itself belongs to
the third-party ``open_clip`` package (open-clip-torch 3.0.0 in the reference's lock file).  It is imported on first
construction; in an environment without it the constructors raise ``ImportError`` like the reference.

What this module adds to the reference's wrappers: ``.native()`` hands the loaded towers to
``NativeClip`` (K11) / ``DevicePreprocess`` (K12) so the same object runs on the package's HIP kernels.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from semanticlens_amd.foundation_models.base import AbstractVLM


@dataclass(frozen=True)
class _Spec:
    """What ``open_clip.create_model_and_transforms`` / ``get_tokenizer`` are asked for."""

    model_name: str
    pretrained: str | None = None


def _load_open_clip(spec: _Spec, device, **kwargs):
    """(model in eval mode on ``device``, inference transform, tokenizer) for ``spec``."""
    import open_clip  # third party; absent -> ImportError, as upstream

    if spec.pretrained is not None:
        kwargs = {"pretrained": spec.pretrained, **kwargs}
    model, _train_transform, transform = open_clip.create_model_and_transforms(spec.model_name, **kwargs)
    return model.eval().to(device), transform, open_clip.get_tokenizer(spec.model_name)


class OpenClip(AbstractVLM):
    """Any model ``open_clip`` can create from ``url`` (clip.py:27-187)."""

    def __init__(self, url: str, device="cpu", **kwargs):
        self._install(_Spec(url), device, kwargs)

    def _install(self, spec: _Spec, device, kwargs):
        self.url = spec.model_name
        self.model, self.preprocessor, self.tokenizer = _load_open_clip(spec, device, **kwargs)

    def __repr__(self):
        return f"{type(self).__name__}(url='{self.url}', model={type(self.model).__name__})"

    # ---- AbstractVLM -------------------------------------------------------------------------------------------
    @property
    def device(self):
        return next(self.model.parameters()).device

    def to(self, device):
        return self.model.to(device)  # the reference returns the torch module here too (clip.py:85-101)

    @torch.no_grad()
    def encode_image(self, img: torch.Tensor) -> torch.Tensor:
        return self.model.encode_image(img)

    @torch.no_grad()
    def encode_text(self, text_input: torch.Tensor) -> torch.Tensor:
        return self.model.encode_text(text_input)

    def preprocess(self, img) -> torch.Tensor:
        """One image or a list of images -> ``(B, 3, S, S)`` on the model's device (clip.py:137-163)."""
        if isinstance(img, list):
            batch = torch.stack([self.preprocessor(sample) for sample in img])
        else:  # a transform that already returns a batch passes through; only a bare (3, S, S) gains the batch axis
            batch = self.preprocessor(img)
        if batch.ndim == 3:
            batch = batch.unsqueeze(0)
        return batch.to(self.device)

    def tokenize(self, txt, context_length: int | None = None) -> torch.Tensor:
        length = context_length if context_length else self.model.context_length
        return self.tokenizer(txt, context_length=length).to(self.device)

    # ---- the package's own execution path --------------------------------------------------------------------------
    #: what ``native()`` does with an image tower the native classes do not read: "native" = raise ``TypeError``
    NATIVE_IMAGE_TOWER = "native"

    def native(self, gemm: str = "bf16x3", device_preprocess: bool = True, image_tower: str | None = None):
        """This model with its towers on the HIP kernels and, optionally, its inference transform on the device
        (``DevicePreprocess.from_transform(self.preprocessor)``).  The native class follows the module layout open_clip
        built: ``NativeClip`` for its own ``VisionTransformer`` and ResNet towers, ``NativeSigLip`` for a timm trunk with a MAP
        head (``SigLipV2``).  ``image_tower="auto"`` (``ClipMobile``'s default: MobileCLIP's FastViT hybrid is not read) keeps
        such an image tower on PyTorch and moves the text tower only (``NativeTextClip``); ``"native"`` raises ``TypeError``."""
        from semanticlens_amd.foundation_models.native_clip import native_model

        return native_model(self, gemm=gemm, preprocess="device" if device_preprocess else None,
                            image_tower=image_tower or self.NATIVE_IMAGE_TOWER)


class SigLipV2(OpenClip):
    """``hf-hub:timm/ViT-B-16-SigLIP2`` (clip.py:190-211)."""

    SPEC = _Spec("hf-hub:timm/ViT-B-16-SigLIP2")
    URL = SPEC.model_name

    def __init__(self, device="cpu", **kwargs):
        self._install(self.SPEC, device, kwargs)


class ClipMobile(OpenClip):
    """MobileCLIP ``s1`` / ``s2`` with the ``datacompdr`` weights (clip.py:214-247)."""

    SPECS = {"s1": _Spec("MobileCLIP-S1", "datacompdr"), "s2": _Spec("MobileCLIP-S2", "datacompdr")}
    URLs = {version: spec.model_name for version, spec in SPECS.items()}
    NATIVE_IMAGE_TOWER = "auto"  # FastViT hybrid image tower: stays on PyTorch; `.native()` moves the text tower (NativeTextClip)

    def __init__(self, version: str = "s1", device="cpu", **kwargs):
        self._install(self.SPECS[version], device, kwargs)
