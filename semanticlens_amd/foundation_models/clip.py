"""OpenCLIP-family wrappers behind ``AbstractVLM`` (reference: foundation_models/clip.py:27-247).

The encoder arithmetic lives in the third-party ``open_clip`` package (open-clip-torch 3.0.0
in the reference's lock file), which this image does not ship: constructing these classes
raises ImportError exactly as the reference does without it.  Any other ``AbstractVLM``
(e.g. ``synth.SyntheticClip`` used by bench.py) plugs into the same seam.
"""
from __future__ import annotations

import torch

from semanticlens_amd.foundation_models.base import AbstractVLM


class OpenClip(AbstractVLM):
    """``open_clip.create_model_and_transforms(url)`` + its tokenizer (clip.py:52-62)."""

    def __init__(self, url, device="cpu", **kwargs):
        import open_clip

        model, _, preprocess = open_clip.create_model_and_transforms(url, **kwargs)
        self.url = url
        self.model = model.eval().to(device)
        self.preprocessor = preprocess
        self.tokenizer = open_clip.get_tokenizer(url)

    def __repr__(self):
        return f"{self.__class__.__name__}(url='{self.url}', model={self.model.__class__.__name__})"

    @property
    def device(self):
        return next(self.model.parameters()).device

    def to(self, device):
        return self.model.to(device)

    def encode_image(self, img: torch.Tensor):
        with torch.no_grad():
            return self.model.encode_image(img)

    def encode_text(self, text_input: torch.Tensor):
        with torch.no_grad():
            return self.model.encode_text(text_input)

    def preprocess(self, img) -> torch.Tensor:
        batch = torch.stack([self.preprocessor(i) for i in img]) if isinstance(img, list) else self.preprocessor(img)
        if batch.ndim == 3:
            batch = batch.unsqueeze(0)
        return batch.to(self.device)

    def tokenize(self, txt, context_length=None):
        context_length = context_length or self.model.context_length
        return self.tokenizer(txt, context_length=context_length).to(self.device)


class SigLipV2(OpenClip):
    """``hf-hub:timm/ViT-B-16-SigLIP2`` (clip.py:190-211)."""

    URL = "hf-hub:timm/ViT-B-16-SigLIP2"

    def __init__(self, device="cpu", **kwargs):
        super().__init__(url=self.URL, device=device, **kwargs)


class ClipMobile(OpenClip):
    """MobileCLIP-S1/S2 with the ``datacompdr`` weights (clip.py:214-247)."""

    URLs = dict(s1="MobileCLIP-S1", s2="MobileCLIP-S2")

    def __init__(self, version="s1", device="cpu", **kwargs):
        import open_clip

        model, _, preprocess = open_clip.create_model_and_transforms(self.URLs[version], pretrained="datacompdr", **kwargs)
        self.model = model.eval().to(device)
        self.url = self.URLs[version]
        self.preprocessor = preprocess
        self.tokenizer = open_clip.get_tokenizer(self.URLs[version])
