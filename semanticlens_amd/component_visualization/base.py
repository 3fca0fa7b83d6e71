"""Plugin seam for component visualizers (reference: component_visualization/base.py:16-183).

``Lens`` only relies on ``caching``, ``storage_dir``, ``metadata`` and
``_compute_concept_db(fm, **kwargs)`` (lens.py:308-329); any user subclass written against the
reference's ABC keeps working here.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch


class AbstractComponentVisualizer(ABC):
    """Finds, per model component, the dataset samples that characterise it."""

    def __init__(self, model: torch.nn.Module, device: str | torch.device | None = None):
        self.model = model
        self.device = device or next(model.parameters()).device
        self.model.to(self.device)

    @abstractmethod
    def run(self, *args, **kwargs) -> None:
        """Process the dataset and cache the per-component evidence (e.g. top-activating samples)."""
        raise NotImplementedError

    @abstractmethod
    def _compute_concept_db(self, cv: AbstractComponentVisualizer, **kwargs) -> dict[str, torch.Tensor]:
        """Return ``{layer: (n_components, n_samples, embed_dim)}``; called by ``Lens`` with the foundation model."""
        raise NotImplementedError

    @abstractmethod
    def get_max_reference(self, layer_name) -> torch.Tensor:
        """``(n_components, n_samples)`` dataset indices of the top samples of a layer."""
        raise NotImplementedError

    def to(self, device: str | torch.device):
        self.device = device
        self.model.to(self.device)
        return self

    @property
    def metadata(self) -> dict[str, str]:
        raise NotImplementedError

    @property
    @abstractmethod
    def caching(self) -> bool:
        raise NotImplementedError

    @property
    @abstractmethod
    def storage_dir(self):
        raise NotImplementedError

    @property
    def device(self):
        return next(self.model.parameters()).device
