"""ActivationComponentVisualizer — collect top-activating samples per component, then embed them.

Mirror of ``semanticlens/component_visualization/activation_based.py`` (reference v0.2.1):
same constructor, properties, cache layout and ``Lens`` contract.  Differences, all inside the
hot loops:

* ``_run`` (activation_based.py:341-358): hooks run the fused device path (K1/K2 + K3); the
  per-layer ``.cpu()`` of the reference's aggregators and the per-batch ``model(...).cpu()``
  are gone, so the loop never synchronises with the host.  ``sample_range`` restricts the pass
  to one shard of the dataset (multi-GPU, see ``semanticlens_amd.distributed``).
* ``_embed_vision_dataset`` (:392-433): embeddings are written into one ``(N, D)`` buffer that
  stays in HBM (1.28 M x 512 x 4 B = 2.6 GB) instead of ``.cpu()`` per batch + ``torch.cat``.
* ``embeds[sample_ids]`` (:387-390) is the K5 gather kernel; ``-1`` ids wrap to the last row
  exactly like torch indexing.
"""
from __future__ import annotations

import logging
import warnings
from pathlib import Path

import torch
from torch import nn
from tqdm import tqdm

from semanticlens_amd import _native as N
from semanticlens_amd.component_visualization import aggregators
from semanticlens_amd.component_visualization._prefetch import PinnedStack, Prefetcher, upload_stage
from semanticlens_amd.component_visualization.activation_caching import ActMaxCache
from semanticlens_amd.component_visualization.base import AbstractComponentVisualizer
from semanticlens_amd.utils.helper import get_fallback_name

logger = logging.getLogger(__name__)


class MissingNameWarning(UserWarning):
    """The model or dataset has no ``.name``; a hash-based fallback names the cache directory."""


class _EmbedStage:
    """State of hot loop 2: the device-resident ``(n_total, D)`` table and, for foundation models that ask for it
    (``fm.embed_accumulate`` images: the native towers, whose GEMMs run at their best rate from ~256 images per call and
    whose embeddings do not depend on the batch they were computed in), preprocessed batches held back until that many
    images are there.  The reference encodes every DataLoader batch on its own (``activation_based.py:392-433``); with its
    default ``batch_size=32`` that is 8 launches of every kernel where one does the same work in 37 % of the time."""

    def __init__(self, fm, n_total: int, batch_hint: int = 0):
        self.fm, self.n_total = fm, int(n_total)
        want = int(getattr(fm, "embed_accumulate", 0) or 0)
        self.target = want if want > max(int(batch_hint), 1) else 0
        self.embeds, self.filled = None, 0
        self._held, self._held_n = [], 0

    def _encode(self, pre):
        out = self.fm.encode_image(pre)
        out = N.to_device(out.detach()).to(torch.float32)
        if self.embeds is None:
            self.embeds = torch.empty((self.n_total, out.shape[1]), dtype=torch.float32, device=out.device)
        self.embeds[self.filled : self.filled + out.shape[0]] = out
        self.filled += out.shape[0]

    def add(self, items, preprocessed=None):
        pre = self.fm.preprocess(items) if preprocessed is None else preprocessed
        if not self.target or not torch.is_tensor(pre):
            self.flush()  # held tensor batches come first: rows are written in dataset order
            return self._encode(pre)
        self._held.append(N.to_device(pre))
        self._held_n += pre.shape[0]
        if self._held_n >= self.target:
            self.flush()

    def flush(self):
        if self._held:
            held, self._held, self._held_n = self._held, [], 0
            self._encode(held[0] if len(held) == 1 else torch.cat(held))

    def finish(self) -> torch.Tensor:
        self.flush()
        if self.embeds is None:
            raise RuntimeError("dataset_fm is empty: nothing to embed")
        assert self.filled == self.n_total, "Number of embeddings does not match number of ids!"
        return self.embeds


class ActivationComponentVisualizer(AbstractComponentVisualizer):
    """Activation-maximisation visualizer (reference: activation_based.py:41-561).

    Parameters are those of the reference plus ``tie_mode`` (``"aten"`` / ``"total"``, see
    ``activation_caching``).
    """

    AGGREGATION_DEFAULTS = {
        "mean": aggregators.aggregate_conv_mean,
        "max": aggregators.aggregate_conv_max,
    }
    #: walk the DataLoaders from a background thread, batches staged in pinned memory and uploaded (or preprocessed)
    #: ahead of the device (``_prefetch.py``); False = the reference's synchronous loops
    prefetch = True

    def __init__(
        self,
        model: nn.Module,
        dataset_model,
        dataset_fm,
        layer_names: list[str],
        num_samples: int,
        device=None,
        aggregate_fn=None,
        cache_dir: str | None = None,
        tie_mode: str | None = None,
    ):
        # NB: like the reference, AbstractComponentVisualizer.__init__ is not called.
        self.model = model
        self.dataset = dataset_model
        self.dataset_fm = dataset_fm
        self._init_cache_dir(cache_dir)
        self._validate_args()

        self.layer_names = layer_names
        self._check_layers()

        device = device or next(model.parameters()).device
        self.model.to(device)

        if aggregate_fn is None:
            logger.warning(f"No aggregation_fn provided using default: {aggregators.aggregate_conv_mean.__name__}")
            aggregate_fn = aggregators.aggregate_conv_mean

        self.actmax_cache = ActMaxCache(
            self.layer_names, n_collect=num_samples, aggregation_fn=aggregate_fn, tie_mode=tie_mode
        )

        if self.caching:
            try:
                self.actmax_cache.load(self.storage_dir)
                logger.info(f"Results loaded from {self.storage_dir}")
            except FileNotFoundError:
                logger.info(f"Results will be stored in {self.storage_dir}")

    # ---- argument / cache-path handling (activation_based.py:187-307) ----------------------------
    def _validate_args(self):
        for obj, what in ((self.model, "Model"), (self.dataset, "Dataset")):
            if hasattr(obj, "name"):
                continue
            name = get_fallback_name(obj)
            if self.caching:
                warnings.warn(
                    f"{what} does not have a name attribute, which is required for reliable caching.\n"
                    f"Using a fallback name: {name}.",
                    MissingNameWarning,
                    stacklevel=3,
                )
            obj.name = name
        if len(self.dataset) != len(self.dataset_fm):
            raise ValueError(
                "Model and foundation model datasets should have the same length.",
                (len(self.dataset), len(self.dataset_fm)),
            )

    def _check_layers(self):
        modules = dict(self.model.named_modules())
        for layer in self.layer_names:
            if layer not in modules:
                raise ValueError(f"Layer '{layer}' not found in model.")

    def _init_cache_dir(self, cache_dir):
        if cache_dir is None:
            logger.warning("No cache dir provided. Results will not be cached!")
            self._cache_root = None
        else:
            self._cache_root = Path(cache_dir)
            self._cache_root.mkdir(parents=True, exist_ok=True)

    @property
    def device(self):
        return next(self.model.parameters()).device

    def to(self, device):
        return self.model.to(device)  # returns the model, like the reference (:258-272)

    @property
    def caching(self) -> bool:
        return self._cache_root is not None

    @property
    def storage_dir(self):
        assert self._cache_root, "No cache dir provided"
        return self._cache_root / self.__class__.__name__ / self.dataset.name / self.model.name

    @property
    def metadata(self) -> dict[str, str]:
        return {**self.actmax_cache.metadata, "dataset": self.dataset.name, "model": self.model.name}

    # ---- hot loop 1: collect (activation_based.py:309-358) ----------------------------------------
    def run(self, batch_size=32, num_workers=0):
        """Top-activating samples per component for every layer; loads the cache when present."""
        if self._cache_root is None:
            return self._run(batch_size=batch_size, num_workers=num_workers)
        try:
            self.actmax_cache.load(self.storage_dir)
            return self.actmax_cache.cache
        except FileNotFoundError:
            logger.debug(f"Activation maximization cache not found at {self.storage_dir}. Running computation...")
            return self._run(batch_size=batch_size, num_workers=num_workers)

    @torch.no_grad()
    def _run(self, batch_size: int = 64, num_workers: int = 0, sample_range: tuple[int, int] | None = None):
        dataset = self.dataset
        if sample_range is not None:  # one shard: ids stay global dataset indices
            start, stop = sample_range
            dataset = torch.utils.data.Subset(self.dataset, range(start, stop))
            for name in self.layer_names:
                self.actmax_cache.sample_idx_counter[name] = start
        dataloader = self._model_loader(dataset, batch_size, num_workers)
        with self.actmax_cache.hook_context(self.model):
            for images in tqdm(dataloader, total=len(dataloader), desc="Collecting ActMax"):
                self.collect_batch(images)

        if self._cache_root and sample_range is None:
            self.actmax_cache.store(self.storage_dir)
            logger.debug(f"Stored activation maximization cache at {self.storage_dir}")
        return self.actmax_cache.cache

    def _model_loader(self, dataset, batch_size, num_workers):
        """Iterable of image batches of ``dataset`` (labels dropped, as the reference's ``for images, _ in ...``)."""
        dev = torch.device(self.device)
        if not (self.prefetch and dev.type == "cuda"):
            loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers)
            return _FirstOfBatch(loader)
        if num_workers == 0:
            stack = PinnedStack()
            loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, collate_fn=stack)
            return Prefetcher(loader, upload_stage(dev, stack), dev)
        loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers,
                                             pin_memory=True)
        return Prefetcher(loader, upload_stage(dev), dev)

    def _fm_loader(self, fm, data, batch_size, collate, **kwargs):
        """Iterable of ``(items, preprocessed or None)`` of ``dataset_fm``: with prefetching ``fm.preprocess`` runs in
        the background thread (device preprocessing: pack + upload + K12 on the thread's stream; a host transform:
        the PIL work of the next batch beside the encoder of this one)."""
        loader = torch.utils.data.DataLoader(data, batch_size=batch_size, shuffle=False, collate_fn=collate, **kwargs)
        dev = torch.device(self.device)
        if not (self.prefetch and dev.type == "cuda"):
            return _WithNone(loader)

        def stage(items):
            pre = fm.preprocess(items)
            if isinstance(pre, torch.Tensor) and not pre.is_cuda:
                pre = pre.to(dev, non_blocking=True)
            return items, pre

        return Prefetcher(loader, stage, dev)

    def collect_batch(self, images: torch.Tensor):
        """One step of hot loop 1: forward ``images`` (host or device resident) under the active hooks.
        Must be called inside ``self.actmax_cache.hook_context(self.model)``; never synchronises."""
        self.model(images.to(self.device, non_blocking=True))

    # ---- hot loop 2 + gather (activation_based.py:360-451) ---------------------------------------
    @torch.no_grad()
    def _compute_concept_db(self, fm, batch_size=32, keep_on_device: bool = False, referenced_only: bool = False,
                            single_pass: bool = False, **kwargs):
        """``{layer: (n_components, n_samples, D)}`` = embeddings of each component's top samples.

        Returns host tensors like the reference unless ``keep_on_device`` is set.  ``referenced_only`` embeds only the
        samples some component actually refers to (at most ``sum(C) * k`` of the ``N`` samples — 72 k of ImageNet's
        1.28 M for ResNet-50 layer2-4 at k=20) instead of the whole dataset as the reference does
        (activation_based.py:392-433); the concept DB is the same whenever ``fm`` embeds a sample independently of
        its batch.  ``single_pass`` walks ``dataset`` and ``dataset_fm`` together, one batch at a
        time: forward + collect on the current HIP stream, the embedding of the same batch on a second stream beside it
        (the reference makes two sequential passes, activation_based.py:341-358 then :392-433); same top-k states and
        concept DB, +6 % throughput in ``bench.py``.  When the top-k cache exists only the embedding pass runs.
        """
        if single_pass and not referenced_only:
            embeds = self._collect_and_embed_single_pass(fm, batch_size, **kwargs)
        else:
            self.run(batch_size=batch_size, **kwargs)
            if referenced_only:
                return self._concept_db_from_referenced(fm, batch_size, keep_on_device, **kwargs)
            embeds = self._embed_vision_dataset(fm, batch_size, **kwargs)
        return self._gather_layers(embeds, {name: self.get_max_reference(name) for name in self.layer_names}, keep_on_device)

    @staticmethod
    def _gather_layers(table: torch.Tensor, refs: dict, keep_on_device: bool) -> dict:
        """``{layer: table[ids]}`` (activation_based.py:387-390) with K5 launched ONCE over the ids of all layers — a layer alone
        is ``C * k * D * 4`` bytes (2-20 MB: launch-latency-sized; one range check and one host synchronisation instead of one
        per layer).  The per-layer results are views of one buffer on the device, separate host tensors otherwise."""
        if not refs:
            return {}
        shapes = [tuple(ids.shape) for ids in refs.values()]
        flat = torch.cat([ids.reshape(-1).to(torch.int64) for ids in refs.values()])
        gathered = N.gather_rows(table, flat)  # (sum C*k, D)
        out, row = {}, 0
        for name, shape in zip(refs, shapes):
            rows = 1
            for d in shape:
                rows *= d
            part = gathered[row : row + rows].reshape(shape + (gathered.shape[1],))
            out[name] = part if keep_on_device else part.cpu()
            row += rows
        return out

    def _collect_and_embed_single_pass(self, fm, batch_size, num_workers: int = 0, **kwargs):
        """Hot loops 1 and 2 fused over one walk of the data, on two HIP streams.  Returns the ``(N, D)`` table."""
        if self._cache_root is not None:
            try:
                self.actmax_cache.load(self.storage_dir)
                return self._embed_vision_dataset(fm, batch_size, num_workers=num_workers, **kwargs)
            except FileNotFoundError:
                logger.debug(f"Activation maximization cache not found at {self.storage_dir}. Running computation...")
        fm.to(self.device)

        def first(batch):
            return [item[0] if isinstance(item, (tuple, list)) else item for item in batch]

        loader_m = self._model_loader(self.dataset, batch_size, num_workers)
        loader_f = self._fm_loader(fm, self.dataset_fm, batch_size, first, num_workers=num_workers)
        n_total = len(self.dataset_fm)
        main = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device)
        stage = _EmbedStage(fm, n_total, batch_size)
        with self.actmax_cache.hook_context(self.model):
            it_f = iter(loader_f)
            for images in tqdm(loader_m, total=len(loader_m), desc="Collecting + embedding"):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    items, pre = next(it_f)
                    stage.add(items, pre)
                self.collect_batch(images)
            for _ in it_f:  # both datasets have the same length (checked by the constructor)
                raise RuntimeError("dataset_fm yielded more batches than dataset")
            with torch.cuda.stream(side):
                embeds = stage.finish()
        main.wait_stream(side)
        if self._cache_root:
            self.actmax_cache.store(self.storage_dir)
        return embeds

    def _concept_db_from_referenced(self, fm, batch_size, keep_on_device, **kwargs):
        n_total = len(self.dataset_fm)
        refs = {name: self.get_max_reference(name) for name in self.layer_names}
        if not refs:
            return {}
        flat = torch.cat([r.reshape(-1) for r in refs.values()]).to(torch.int64)
        # the -1 sentinel of a never-filled slot gathers the LAST embedding, like `embeds[-1]`
        flat = torch.where(flat < 0, flat + n_total, flat)
        if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= n_total):
            raise IndexError(f"index out of range in embeds[sample_ids] (dataset size {n_total})")
        uniq = torch.unique(flat)  # sorted
        table = self._embed_vision_dataset(fm, batch_size, subset=uniq.tolist(), **kwargs)
        pos = {}
        for name, ids in refs.items():
            ids = ids.to(torch.int64)
            pos[name] = torch.searchsorted(uniq, torch.where(ids < 0, ids + n_total, ids))
        return self._gather_layers(table, pos, keep_on_device)

    def _embed_vision_dataset(self, fm, batch_size, subset=None, **kwargs):
        """Embed every ``dataset_fm`` sample (or the samples listed in ``subset``, in that order) with ``fm``; returns
        the ``(N, D)`` fp32 table resident on the device."""
        fm.to(self.device)

        def pil_list_collate(batch):
            if isinstance(batch[0], (tuple, list)):
                return [item[0] for item in batch]
            return list(batch)

        data = self.dataset_fm if subset is None else torch.utils.data.Subset(self.dataset_fm, subset)
        loader = self._fm_loader(fm, data, batch_size, pil_list_collate, **kwargs)
        n_total = len(data)
        stage = _EmbedStage(fm, n_total, batch_size)
        with tqdm(total=n_total, desc="Embedding Dataset") as pbar:
            for items, pre in loader:
                stage.add(items, pre)
                pbar.update(batch_size)
        return stage.finish()

    @staticmethod
    def embed_batch(fm, items, embeds, filled: int, n_total: int, preprocessed=None):
        """One step of hot loop 2: ``fm.encode_image(fm.preprocess(items))`` written into rows
        ``[filled, filled + B)`` of the device-resident ``(n_total, D)`` table (allocated on first use).
        ``preprocessed`` = the result of ``fm.preprocess(items)`` when the loader already produced it."""
        out = fm.encode_image(fm.preprocess(items) if preprocessed is None else preprocessed)
        out = N.to_device(out.detach()).to(torch.float32)
        if embeds is None:
            embeds = torch.empty((n_total, out.shape[1]), dtype=torch.float32, device=out.device)
        embeds[filled : filled + out.shape[0]] = out
        return embeds, filled + out.shape[0]

    def get_max_reference(self, layer_name) -> torch.Tensor:
        """``(n_components, n_samples)`` int64 dataset indices (``-1`` = slot never filled)."""
        self._check_layer_name(layer_name)
        return self.actmax_cache.cache[layer_name].sample_ids

    def visualize_components(self, component_ids, layer_name: str, n_samples: int = 9, nrows: int = 3, fname=None,
                             denormalization_fn=None):
        """Plot the top activating samples of some components (reference: activation_based.py:453-543; host-side
        matplotlib rendering, not part of the device path).  One tile per component: its first ``n_samples``
        reference samples from ``self.dataset`` laid out ``nrows`` per row (the reference passes ``nrows`` as
        ``make_grid``'s images-per-row), tiles arranged in a near-square figure; saved under
        ``<storage_dir>/plots`` when caching is on."""
        self._check_layer_name(layer_name)
        import matplotlib.pyplot as plt

        post = getattr(self.dataset, "denormalization_fn", None) or denormalization_fn or (lambda x: x)
        ids_all = self.get_max_reference(layer_name=layer_name)
        component_ids = torch.as_tensor(component_ids).reshape(-1)
        tiles = []
        for cid in component_ids.tolist():
            imgs = [torch.as_tensor(post(self.dataset[int(i)][0])).detach().cpu() for i in ids_all[cid][:n_samples]]
            tiles.append(_image_grid(imgs, per_row=nrows).permute(1, 2, 0).numpy())
        n = len(tiles)
        cols = max(1, int(n**0.5))
        rows = (n + cols - 1) // cols
        fig, axs = plt.subplots(rows, cols, figsize=(3 * cols, 3 * rows), squeeze=False)
        flat = axs.reshape(-1)
        for ax, tile, cid in zip(flat, tiles, component_ids.tolist()):
            ax.imshow(tile)
            ax.set_title(f"Neuron {cid}")
            ax.set_xticks([])
            ax.set_yticks([])
        for ax in flat[n:]:
            ax.axis("off")
        model_name = str(getattr(self.model, "name", type(self.model).__name__))
        plt.suptitle((f"{fname:.15} " if fname else "") + f"{model_name:>.10} {layer_name:<.15}", fontsize=16)
        plt.tight_layout(rect=[0, 0, 1, 0.96])
        plt.show()
        if self.caching:
            fdir = self.storage_dir / "plots"
            fdir.mkdir(parents=True, exist_ok=True)
            tag = "-".join(str(c) for c in component_ids.tolist())
            fpath = fdir / ((fname + "_" if fname else "") + f"{layer_name}_{tag}.png")
            plt.savefig(fpath)
            plt.close(fig)
            print(f"Saved visualization to {fpath}")
        elif fname:
            logger.warning(
                "Failed to save visualization. Caching is not enabled in the ComponentVisualizer (`cv.caching: False`)"
            )
        return fig

    def _check_layer_name(self, layer_name):
        if layer_name not in self.layer_names:
            raise ValueError(f"Layer '{layer_name}' not found in model layers: {self.layer_names}")


class _FirstOfBatch:
    """``(images, labels)`` batches -> ``images``."""

    def __init__(self, loader):
        self.loader = loader

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            yield batch[0] if isinstance(batch, (tuple, list)) else batch


class _WithNone:
    """``items`` -> ``(items, None)``: the synchronous form of ``_fm_loader``."""

    def __init__(self, loader):
        self.loader = loader

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for items in self.loader:
            yield items, None


def _image_grid(images: list[torch.Tensor], per_row: int, padding: int = 2) -> torch.Tensor:
    """``(3, H', W')`` mosaic of equally sized ``(C, H, W)`` images, ``per_row`` per row with a ``padding``-pixel black
    frame around every cell — the layout ``torchvision.utils.make_grid`` produces (torchvision is not a dependency)."""
    if not images:
        raise ValueError("no images to arrange")
    imgs = []
    for im in images:
        im = im.float()
        if im.ndim == 2:
            im = im[None]
        if im.shape[0] == 1:
            im = im.expand(3, -1, -1)
        imgs.append(im)
    C, H, W = imgs[0].shape
    xm = min(per_row, len(imgs))
    ym = (len(imgs) + xm - 1) // xm
    grid = torch.zeros((C, ym * (H + padding) + padding, xm * (W + padding) + padding))
    for idx, im in enumerate(imgs):
        r, c = divmod(idx, xm)
        y0, x0 = r * (H + padding) + padding, c * (W + padding) + padding
        grid[:, y0 : y0 + H, x0 : x0 + W] = im
    return grid
