"""Plugin seam for vision-language foundation models (reference: foundation_models/base.py:12-120).

Any object with these members works with ``Lens`` and ``ActivationComponentVisualizer``;
features are returned un-normalised (clip.py:117-118,134-135) and normalised by the scores.
"""
from abc import ABC, abstractmethod


class AbstractVLM(ABC):
    @abstractmethod
    def encode_image(self, *args, **kwargs):
        """Preprocessed image batch -> (B, D) features."""

    @abstractmethod
    def encode_text(self, *args, **kwargs):
        """Token batch -> (B, D) features."""

    @abstractmethod
    def preprocess(self, img):
        """PIL image or list of PIL images -> model-ready batch on ``self.device``."""

    @abstractmethod
    def tokenize(self, txt: str):
        """String or list of strings -> token tensor."""

    @property
    @abstractmethod
    def device(self):
        """Device of the model parameters."""

    @abstractmethod
    def to(self, device):
        """Move the model to ``device``."""
