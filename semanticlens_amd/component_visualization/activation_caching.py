"""Streaming top-k activation collection (ActMax / ActCache / ActMaxCache) on the MI355X.

Mirror of ``semanticlens/component_visualization/activation_caching.py`` (reference v0.2.1):
same classes, attributes, cache-file layout and error behaviour.  What differs is where the
work happens.  The reference aggregates on the device, copies a ``(B, C)`` tensor to the host
per layer per batch and runs ``cat -> torch.topk -> gather`` on the CPU
(activation_caching.py:133-141).  Here the top-k state ``(C, k)`` bf16 + int64 lives in HBM and
a forward hook launches two kernels on torch's current stream, with no host synchronisation:

* K1/K2 (``csrc/reduce.hip``): activation -> ``(B, C)`` bf16 candidates,
* K3 (``csrc/actmax.hip`` / ``actmax_aten.hip``): merge candidates into the state.

Tie modes (``tie_mode`` argument, default from ``$SEMANTICLENS_AMD_TIES``, else ``"aten"``):

``"aten"``   bit-identical to the reference: one merge per batch reproducing torch.topk's CPU
             tie order on ``cat([state, batch])``.
``"total"``  value descending then sample id ascending.  Independent of batch size and of how
             the dataset is sharded (used for multi-GPU); candidate batches are queued in a ring
             and merged every ``$SEMANTICLENS_AMD_MERGE_EVERY`` (default 8) batches.
Both store the same bf16 values; they can differ only in which of several equal-valued samples
is listed (see DESIGN.md "Tie semantics").
"""
from __future__ import annotations

import inspect
import logging
import os
import warnings
from collections import Counter, OrderedDict
from collections.abc import Callable
from contextlib import contextmanager
from pathlib import Path
from typing import Any

import safetensors
import safetensors.torch
import torch

from semanticlens_amd import _native as N

from . import aggregators

logger = logging.getLogger(__name__)

DEFAULT_AGGREGATION_FUNCTION_MAP = {name: func for name, func in inspect.getmembers(aggregators, inspect.isfunction)}


def default_tie_mode() -> str:
    mode = os.environ.get("SEMANTICLENS_AMD_TIES", "aten").lower()
    if mode not in N.TIE_MODES:
        raise ValueError(f"SEMANTICLENS_AMD_TIES must be one of {sorted(N.TIE_MODES)}, got {mode!r}")
    return mode


def _merge_every() -> int:
    return max(1, min(N.SL_MAX_SLOTS, int(os.environ.get("SEMANTICLENS_AMD_MERGE_EVERY", "8"))))


class ActMax:
    """Top-``n_collect`` activations (bf16) and their sample ids (int64) per latent.

    Reference: activation_caching.py:64-216.  ``activations`` / ``sample_ids`` are host tensors
    as in the reference; reading them flushes pending device work and copies the state down.
    """

    def __init__(self, n_collect: int, n_latents: int | None = None, tie_mode: str | None = None,
                 init_value: float = -0.0):
        self.n_collect = n_collect
        self.n_latents = n_latents
        self.is_setup = False
        # the reference starts every slot at -0.0 (fine for post-ReLU activations); signed quantities (relevance) start
        # at -inf so that an empty slot (id -1) never outranks a real negative value
        self.init_value = float(init_value)
        self.tie_mode = tie_mode or default_tie_mode()
        if self.tie_mode not in N.TIE_MODES:
            raise ValueError(f"tie_mode must be one of {sorted(N.TIE_MODES)}, got {self.tie_mode!r}")
        self._host_vals: torch.Tensor | None = None
        self._host_ids: torch.Tensor | None = None
        self._dev_vals: torch.Tensor | None = None  # (C,k) bf16 in HBM
        self._dev_ids: torch.Tensor | None = None  # (C,k) int64 in HBM
        self._dev_newer = False
        self._ring: torch.Tensor | None = None  # (slots, Bcap, C) bf16 candidate batches awaiting a merge
        self._policy_tuner = None  # N.ReducePolicyTuner of this layer's reduce (created on the first fused collect)
        self._pending: list[tuple[int, int]] = []  # (id_base, rows) per queued slot
        self._ws: torch.Tensor | None = None
        self._before_flush = None  # set by ActMaxCache for layers collected in groups: runs the group's queued work first
        if n_latents is not None:
            self._setup_tensors()

    # ---- state -------------------------------------------------------------------------------
    def _setup_tensors(self):
        """Initial state: values -0.0 (``init_value``), ids -1 (activation_caching.py:101-110)."""
        self._host_vals = torch.full((self.n_latents, self.n_collect), self.init_value, dtype=torch.bfloat16)
        self._host_ids = -torch.ones(self.n_latents, self.n_collect, dtype=torch.int64)
        self._dev_vals = self._dev_ids = None
        self._dev_newer = False
        self.is_setup = True

    def _device_state(self, device: torch.device):
        if self._dev_vals is None or self._dev_vals.device != device:
            if self.tie_mode == "aten":
                # first use of the reference's tie order in this process: the restatement the kernels evaluate is checked against the
                # installed torch.topk (200 tie-heavy rows, ~10 ms of host work, cached); a differing host is warned about
                N.aten_order_selftest()
            self.flush()
            self._sync_host()
            self._dev_vals = self._host_vals.to(device).contiguous()
            self._dev_ids = self._host_ids.to(device).contiguous()
            self._ring = None
        return self._dev_vals, self._dev_ids

    def _sync_host(self):
        if self._dev_newer:
            self._host_vals = self._dev_vals.cpu()
            self._host_ids = self._dev_ids.cpu()
            self._dev_newer = False

    @property
    def activations(self) -> torch.Tensor:
        self.flush()
        self._sync_host()
        return self._host_vals

    @activations.setter
    def activations(self, value: torch.Tensor):
        self.flush()
        self._sync_host()
        self._host_vals = value.detach().to("cpu", torch.bfloat16)
        self._dev_vals = self._dev_ids = None

    @property
    def sample_ids(self) -> torch.Tensor:
        self.flush()
        self._sync_host()
        return self._host_ids

    @sample_ids.setter
    def sample_ids(self, value: torch.Tensor):
        self.flush()
        self._sync_host()
        self._host_ids = value.detach().to("cpu", torch.int64)
        self._dev_vals = self._dev_ids = None

    def device_state(self, device=None):
        """(values bf16, ids int64) resident on the HIP device, pending merges applied."""
        device = torch.device(device) if device is not None else (
            self._dev_vals.device if self._dev_vals is not None else N.default_device()
        )
        state = self._device_state(device)
        self.flush()
        return state

    # ---- updates -----------------------------------------------------------------------------
    def update(self, acts: torch.Tensor, sample_ids: torch.Tensor):
        """Merge a ``(B, n_latents)`` batch with explicit sample ids (activation_caching.py:112-141)."""
        assert acts.ndim == 2
        if not self.is_setup:
            self.n_latents = acts.shape[1]
            self._setup_tensors()
        a = N.to_device(acts.detach())
        B, C = a.shape
        if C != self.n_latents:
            raise RuntimeError(f"activation batch has {C} latents, ActMax was set up for {self.n_latents}")
        if self.n_collect == 0 or B == 0:
            return
        vals, ids = self._device_state(a.device)
        self.flush()
        cand = torch.empty((B, C), dtype=torch.bfloat16, device=a.device)
        # (B, C) -> bf16 exactly as `acts.T.to(bfloat16)` (:133): K1 with a 1-element reduction axis
        N.reduce_conv(a.reshape(B, C, 1, 1), N.SL_CONV_MAX, cand, None)
        sid = N.to_device(sample_ids.detach(), a.device).to(torch.int64).contiguous()
        if sid.shape != (B,):
            raise RuntimeError(f"sample_ids has shape {tuple(sid.shape)}, expected ({B},)")
        N.actmax_update(vals, ids, cand, sid, 0, B, N.TIE_MODES[self.tie_mode], self._aten_ws(B, a.device))
        self._dev_newer = True

    def collect(self, outs: torch.Tensor, native: tuple, id_base: int, site=None, k3_queue=None):
        """Fused hook path: reduce ``outs`` on the device and merge; samples get ids id_base + b.  ``site``: what identifies the
        producer of ``outs`` (the hooked module), for the cache-policy tuner of its reduce.  ``k3_queue(act_max, cand, id_base, B)``
        (``ActMaxCache``, reference tie order only): takes the merge over, to run it with the other layers' of this forward."""
        kind, code, pos = native
        x = N.to_device(outs.detach())
        B = x.shape[0]
        C = x.shape[1] if kind == "conv" else x.shape[2]
        if not self.is_setup:
            self.n_latents = C
            self._setup_tensors()
        if C != self.n_latents:
            raise RuntimeError(f"layer output has {C} components, ActMax was set up for {self.n_latents}")
        if self.n_collect == 0 or B == 0:
            return
        vals, ids = self._device_state(x.device)
        slots = 1 if self.tie_mode == "aten" else _merge_every()
        if self._ring is None or self._ring.shape[1] < B or self._ring.shape[0] != slots:
            self.flush()
            self._ring = torch.empty((slots, B, C), dtype=torch.bfloat16, device=x.device)
        slot = self._ring[len(self._pending)]
        cand = slot[:B]
        if self._policy_tuner is None:
            self._policy_tuner = N.ReducePolicyTuner.for_site(site) if site is not None else N.ReducePolicyTuner()
        if kind == "conv":
            self._policy_tuner.run(lambda: N.reduce_conv(x, code, cand, None), x.numel() * x.element_size(), B)
        else:
            self._policy_tuner.run(lambda: N.reduce_tokens(x, code, pos, cand, None), x.numel() * x.element_size(), B)
        if self.tie_mode == "aten":
            if k3_queue is not None:
                k3_queue(self, cand, id_base, B)
                return
            N.actmax_update(vals, ids, cand, None, id_base, B, N.SL_TIES_ATEN, self._aten_ws(B, x.device))
            self._dev_newer = True
        else:
            self._pending.append((id_base, B))
            if len(self._pending) == slots:
                self.flush()

    def flush(self):
        """Merge every queued candidate batch into the state (total-order mode); a layer collected as part of a group of
        identical layers (``ActMaxCache``) first has its group's stashed activations reduced and merged."""
        if self._before_flush is not None:
            self._before_flush()
        if self._pending:
            ring = self._ring
            N.actmax_merge(
                self._dev_vals, self._dev_ids, ring, ring.stride(0), [p[0] for p in self._pending],
                [p[1] for p in self._pending],
            )
            self._pending = []
            self._dev_newer = True

    def merge_states(self, other_vals: torch.Tensor, other_ids: torch.Tensor):
        """K4: fold ``R`` other states ``(R, C, k)`` (e.g. all-gathered ranks) into this one, total order."""
        if not self.is_setup:
            raise RuntimeError("merge_states on an ActMax that is not set up")
        ov = N.to_device(other_vals)
        vals, ids = self._device_state(ov.device)
        self.flush()
        oi = N.to_device(other_ids, ov.device).to(torch.int64).contiguous()
        ov = ov.to(torch.bfloat16).contiguous()
        if ov.shape[1:] != vals.shape or oi.shape != ov.shape:
            raise RuntimeError(f"states to merge have shape {tuple(ov.shape)}, expected (R, {vals.shape[0]}, {vals.shape[1]})")
        N.actmax_merge_states(vals, ids, ov, oi)
        self._dev_newer = True

    def _aten_ws(self, B: int, device):
        if self.tie_mode != "aten":
            return None
        need = N.actmax_aten_ws_bytes(self.n_latents, self.n_collect, B)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    # ---- reference utilities -------------------------------------------------------------------
    @property
    def alive_latents(self) -> torch.Tensor:
        """Indices of latents with any non-zero stored activation (activation_caching.py:143-156)."""
        if not self.is_setup:
            return torch.tensor([], dtype=torch.int64)
        return torch.where(self.activations.abs().sum(dim=1) > 0)[0]

    def store(self, file_path: str | Path, metadata: dict[str, str] | None = None):
        """Write ``activations`` / ``sample_ids`` + metadata as safetensors (activation_caching.py:158-182)."""
        if not self.is_setup:
            logger.warning("Attempted to store an un-initialized ActMax instance; skipping.")
            return
        tensors = {"activations": self.activations, "sample_ids": self.sample_ids}
        safetensors.torch.save_file(tensors, file_path, metadata=metadata)
        logger.debug(f"Stored ActMax data to {file_path}")

    @classmethod
    def load(cls, file_path: str | Path) -> ActMax:
        """Read a file written by :meth:`store` (or by the reference; activation_caching.py:184-216)."""
        with safetensors.safe_open(file_path, framework="pt") as f:
            metadata = f.metadata()
            if metadata is None:
                raise ValueError(f"File {file_path} is missing required metadata for loading.")
            tensors = {k: f.get_tensor(k) for k in f.keys()}
        instance = cls(n_collect=int(metadata["n_collect"]), n_latents=int(metadata["n_latents"]))
        instance.activations = tensors["activations"]
        instance.sample_ids = tensors["sample_ids"]
        return instance


class ActCache:
    """Forward-hook plumbing: keeps each hooked layer's raw output (activation_caching.py:219-315)."""

    def __init__(self, layer_names: list[str]):
        self.layer_names = layer_names
        self.cache: dict[str, Any] = OrderedDict()
        self.handles: list[torch.utils.hooks.RemovableHandle] = []

    def _get_hook(self, name: str) -> Callable:
        def hook_fn(module, ins, outs):
            self.cache[name] = outs.detach().cpu()

        return hook_fn

    def _register_hooks(self, model: torch.nn.Module):
        for name, module in model.named_modules():
            if name in self.layer_names:
                self.handles.append(module.register_forward_hook(self._get_hook(name)))

    def _finalize(self):
        pass

    @contextmanager
    def hook_context(self, model: torch.nn.Module):
        """Register the hooks for the duration of the ``with`` block; always removes them."""
        self._register_hooks(model)
        try:
            yield
        finally:
            for handle in self.handles:
                handle.remove()
            self.handles.clear()
            self._finalize()


class ActMaxCache(ActCache):
    """Per-layer :class:`ActMax` instances fed by forward hooks (activation_caching.py:318-534)."""

    def __init__(self, layer_names: list[str], aggregation_fn: Callable, n_collect: int, tie_mode: str | None = None,
                 init_value: float = -0.0):
        super().__init__(layer_names)
        self.aggregation_fn = aggregation_fn
        self.n_collect = n_collect
        self.init_value = float(init_value)
        self.tie_mode = tie_mode or default_tie_mode()
        self.sample_idx_counter = Counter()  # per layer, never reset (activation_caching.py:359,410-413)

        agg_fn_name = getattr(self.aggregation_fn, "__name__", None)
        if agg_fn_name is None or agg_fn_name == "<lambda>":
            raise ValueError("Aggregation function must be a defined function, not a lambda.")
        self.agg_fn_name = agg_fn_name

        self.cache: dict[str, ActMax] = {
            name: ActMax(n_collect=n_collect, tie_mode=self.tie_mode, init_value=self.init_value) for name in layer_names
        }
        # ---- groups of identical layers (round 5) ----
        # Hooking "all encoder blocks" of a transformer means L outputs of ONE shape per batch, each a launch of ~150 MB for K2
        # and a 17 us K3 launch: both sit on their fixed costs.  Layers whose outputs agree in shape, strides, dtype and aggregator
        # are therefore collected TOGETHER: the hook only stashes a reference, and when the last member of the group has fired
        # one `sl_reduce_*_multi` launch reads all L tensors (1.9 GB for ViT-B/16's twelve blocks at B = 256) and one
        # `sl_actmax_update_multi` launch merges all L states.  Costs L - 1 activations kept alive a little longer (HBM is 288 GB).
        # Safety: a stashed tensor must not change between its hook and the group's launch.  The first batch runs ungrouped and
        # records every hooked output's `_version`; only layers whose output was still unmodified when the next batch began are
        # grouped, and every later launch re-checks the versions (an in-place edit raises instead of collecting wrong values).
        # The reference's tie order only (`tie_mode="aten"`); SEMANTICLENS_AMD_GROUP_LAYERS=0 switches it off.
        # A version mismatch at launch time (an in-place edit that the first batch did not show) takes that layer out of its group
        # for the rest of the run, with one warning; SEMANTICLENS_AMD_GROUP_LAYERS=strict raises instead.
        self._grouping = self.tie_mode == "aten" and os.environ.get("SEMANTICLENS_AMD_GROUP_LAYERS", "1") != "0"
        self._group_strict = os.environ.get("SEMANTICLENS_AMD_GROUP_LAYERS", "1") == "strict"
        self._group_device_types = ("cuda",)  # host tensors only in tests/test_layer_groups_host.py (kernels replaced by stand-ins)
        # ---- one top-k merge per forward (round 5; opt-in: SEMANTICLENS_AMD_BATCH_K3=1) ----
        # K3's time is a chain of LDS round trips per row, not a function of the number of rows, so the merges of ALL hooked layers
        # of a forward (any shapes: each layer keeps its own candidate matrix) can wait for the last layer and run as ONE
        # `sl_actmax_update_multi` launch: ResNet-50 layer2-4 3 x 18-20 us -> ~23, K1 + K3 per batch 175 -> 139 us.  No activation is
        # kept for this (K1 / K2 ran in the hook); the queue is flushed when the last layer of the first batch's firing order has
        # fired, when a layer fires again, on any state read and when the hooks are removed.  Off by default: end to end nothing
        # moves (the merges are 0.1 % of a step), and with the embed stage on a second stream the shifted phase between the two
        # streams cost K1 0.011 of its in-pipeline fraction in every A/B pair (profiles/r05_k3_batching_ab.txt; none on one stream).
        self._batch_k3 = self._grouping and os.environ.get("SEMANTICLENS_AMD_BATCH_K3", "0") == "1"
        self._k3_queue: list[tuple] = []  # (layer name, ActMax, candidates (B, C), id_base, B)
        self._k3_last: str | None = None  # the last layer of a forward, known once the first batch is over
        self._probe: dict[str, tuple] | None = {}  # first batch: layer -> (signature, tensor, version); None once planned
        self._group_of: dict[str, int] = {}
        self._groups: list[dict] = []

    def __getitem__(self, layer_name: str) -> ActMax:
        return self.cache[layer_name]

    def _register_hooks(self, model: torch.nn.Module):
        super()._register_hooks(model)
        if self._grouping:
            # a forward BOUNDARY (round 6, advisor): groups are planned when the first forward is over — every in-place edit of that
            # forward has bumped its tensor's version by then — and nothing stashed or queued outlives the forward that produced it
            self.handles.append(model.register_forward_hook(self._end_of_forward))

    def _end_of_forward(self, module, ins, outs):
        if self._probe is not None:
            if self._probe:
                self._plan_groups()  # also drops the first batch's tensor references
            return
        self._flush_deferred()  # a forward that did not reach every member of a group; queued merges

    def __iter__(self):
        return iter(self.cache.values())

    def _get_hook(self, layer_name: str) -> Callable:
        native = getattr(self.aggregation_fn, "_sl_native", None)

        def hook_fn(module, ins, outs):
            start = self.sample_idx_counter[layer_name]
            if native is not None:
                # same argument check (and error text) the aggregator itself makes
                want = 4 if native[0] == "conv" else 3
                if outs.ndim != want:
                    raise ValueError(f"Input tensor should be {want}D. \n" + aggregators._ERROR_MESSAGE)
                batch_size = outs.shape[0]
                self.sample_idx_counter[layer_name] += batch_size
                if self._grouping and self._collect_grouped(layer_name, module, outs, native, start):
                    return
                queue = None
                if self._batch_k3 and self._k3_last is not None:
                    self._flush_k3(only_if_queued=layer_name)  # its candidate buffer is about to be rewritten
                    queue = (lambda am, cand, base, B, name=layer_name: self._enqueue_k3(name, am, cand, base, B))
                self.cache[layer_name].collect(outs, native, start, site=(id(module), layer_name), k3_queue=queue)
                return
            # user-defined aggregator: (B, C) tensor on any device, then K3 alone
            aggregated_acts = self.aggregation_fn(outs)
            batch_size = aggregated_acts.shape[0]
            assert aggregated_acts.ndim == 2, "Something is wrong with the aggregation_fn"
            sample_ids = torch.arange(start, start + batch_size)
            self.sample_idx_counter[layer_name] += batch_size
            self.cache[layer_name].update(aggregated_acts, sample_ids)

        return hook_fn

    # ---- groups of identical layers --------------------------------------------------------------------------------------
    @staticmethod
    def _signature(outs: torch.Tensor, native) -> tuple:
        return (native, outs.dtype, tuple(outs.shape[1:]), tuple(outs.stride()), outs.device)

    @staticmethod
    def _version(tensor: torch.Tensor):
        try:
            return tensor._version
        except RuntimeError:  # inference tensors keep no version counter: an in-place edit could not be noticed
            return None

    def _plan_groups(self):
        """End of the first pass over the hooked layers: group the layers with identical outputs that nobody modified in place."""
        by_sig: dict[tuple, list[str]] = {}
        for name, (sig, tensor, version) in self._probe.items():
            if version is not None and self._version(tensor) == version:
                # the batch stride scales with B (a short last batch) and is re-checked per launch; the inner strides (memory
                # format) must agree for two layers to share a group
                by_sig.setdefault(sig[:3] + (sig[3][1:],) + sig[4:], []).append(name)
        if self._batch_k3 and self._probe:
            self._k3_last = list(self._probe)[-1]
            for name in self._probe:
                self.cache[name]._before_flush = self._flush_deferred
        self._probe = None
        for names in by_sig.values():
            if len(names) >= 2:
                gid = len(self._groups)
                self._groups.append({"layers": names, "stash": {}, "cand": {}})
                for name in names:
                    self._group_of[name] = gid
                    self.cache[name]._before_flush = self._flush_deferred

    def _collect_grouped(self, layer_name, module, outs, native, start) -> bool:
        """True when ``outs`` was taken over (stashed, or collected with its group); False = collect it now, alone."""
        if self._probe is not None:
            if layer_name not in self._probe:
                if outs.device.type in self._group_device_types:
                    self._probe[layer_name] = (self._signature(outs, native), outs.detach(), self._version(outs))
                return False
            self._plan_groups()  # a layer fires for the second time: the first batch is over
        gid = self._group_of.get(layer_name)
        version = self._version(outs) if gid is not None else None
        if version is None:
            return False
        group = self._groups[gid]
        self.cache[layer_name]._before_flush = self._flush_deferred  # the ActMax may have been replaced (load)
        if layer_name in group["stash"]:  # the previous forward did not reach every member: finish it layer by layer
            self._flush_group(gid)
        if group["stash"]:
            first = next(iter(group["stash"].values()))
            if self._signature(first[0], first[3]) != self._signature(outs, native) or first[0].shape[0] != outs.shape[0]:
                self._flush_group(gid)
                return False
        stash = group["stash"]  # (_flush_group installs a fresh dict)
        stash[layer_name] = (outs.detach(), version, start, native, module)
        if all(name in stash for name in group["layers"]):
            self._run_group(gid)
        return True

    def _check_unmodified(self, layer_name, tensor, version, start=None) -> bool:
        """False when ``tensor`` was written in place since its hook stashed it.  The layer then leaves its group for good (it is
        collected inside its own hook from the next batch on, like the reference does) and ONE warning says which samples were
        aggregated from the edited tensor; ``SEMANTICLENS_AMD_GROUP_LAYERS=strict`` raises instead."""
        if self._version(tensor) == version:
            return True
        if self._group_strict:
            raise RuntimeError(
                f"the output of hooked layer {layer_name!r} was modified in place after its forward hook ran; layers with identical "
                "outputs are collected together once the last of them has fired, which needs the earlier outputs intact. "
                "Set SEMANTICLENS_AMD_GROUP_LAYERS=0 to collect every layer inside its own hook.")
        self._ungroup(layer_name)
        where = "" if start is None else f" (samples {start}..{start + tensor.shape[0] - 1} of this layer were aggregated AFTER the edit)"
        warnings.warn(
            f"the output of hooked layer {layer_name!r} was modified in place after its forward hook ran{where}; the layer is collected "
            "inside its own hook from now on.  Set SEMANTICLENS_AMD_GROUP_LAYERS=0 to collect every layer that way from the first "
            "batch, or =strict to raise here.", RuntimeWarning, stacklevel=2)
        return False

    def _ungroup(self, layer_name: str):
        """Take ``layer_name`` out of its group; a group left with one member is dissolved."""
        gid = self._group_of.pop(layer_name, None)
        if gid is None:
            return
        group = self._groups[gid]
        group["layers"] = [n for n in group["layers"] if n != layer_name]
        if any(q[0] in group["layers"] or q[0] == layer_name for q in self._k3_queue):
            self._flush_k3()  # queued merges still read the (L, B, C) candidate buffers that are dropped next
        group["cand"] = {}  # sized for the old member count
        if len(group["layers"]) < 2:
            for n in group["layers"]:
                self._group_of.pop(n, None)
            group["layers"] = []

    def _run_group(self, gid: int):
        """All members have fired: one multi-tensor reduce, one multi-state top-k update."""
        group = self._groups[gid]
        stash, names = group["stash"], group["layers"]
        names = list(names)
        entries = [stash[name] for name in names]
        group["stash"] = {}
        intact = [self._check_unmodified(name, tensor, version, start) for name, (tensor, version, start, _, _) in zip(names, entries)]
        if not all(intact):  # nothing of this batch is lost: every member is collected now, alone
            for name, (tensor, _, start, nat, module) in zip(names, entries):
                am = self.cache[name]
                hook, am._before_flush = am._before_flush, None
                try:
                    am.collect(tensor, nat, start, site=(id(module), name))
                finally:
                    am._before_flush = hook
            return
        x0, _, _, native, _ = entries[0]
        kind, code, pos = native
        B = x0.shape[0]
        C = x0.shape[1] if kind == "conv" else x0.shape[2]
        states = []
        for name in names:
            am = self.cache[name]
            if not am.is_setup:
                am.n_latents = C
                am._setup_tensors()
            if C != am.n_latents:
                raise RuntimeError(f"layer output has {C} components, ActMax was set up for {am.n_latents}")
            hook, am._before_flush = am._before_flush, None  # _device_state flushes: nothing of this group is pending any more
            try:
                states.append(am._device_state(x0.device))
            finally:
                am._before_flush = hook
        if self.n_collect == 0 or B == 0:
            return
        if not N.actmax_update_multi_supported(C, self.n_collect, B):
            for name, (tensor, _, start, nat, module) in zip(names, entries):
                self.cache[name].collect(tensor, nat, start, site=(id(module), name))
            return
        cand = group["cand"].get(B)
        if cand is None or cand.device != x0.device or cand.shape[0] != len(names):  # (a member may have left the group)
            cand = group["cand"][B] = torch.empty((len(names), B, C), dtype=torch.bfloat16, device=x0.device)
        if any(q[0] in names for q in self._k3_queue):  # merges of an earlier forward still read this candidate buffer
            self._flush_k3()
        N.reduce_multi(kind, [e[0] for e in entries], code, pos, cand)
        if self._batch_k3 and self._k3_last is not None:
            for l, (name, e) in enumerate(zip(names, entries)):
                self._enqueue_k3(name, self.cache[name], cand[l], e[2], B)
            return
        N.actmax_update_multi(states, [cand[l] for l in range(len(names))], [e[2] for e in entries], B)
        for name in names:
            self.cache[name]._dev_newer = True

    # ---- one top-k merge per forward ----------------------------------------------------------------------------------------
    def _enqueue_k3(self, name, am, cand, id_base, B):
        if any(q[0] == name for q in self._k3_queue) or (self._k3_queue and self._k3_queue[0][4] != B):
            self._flush_k3()
        self._k3_queue.append((name, am, cand, id_base, B))
        stashed = sum(len(g["stash"]) for g in self._groups)
        if name == self._k3_last and not stashed:
            self._flush_k3()

    def _flush_k3(self, only_if_queued: str | None = None):
        """Run the queued merges: one launch for all of them (or layer by layer when k + B rows do not fit the wave kernel)."""
        if not self._k3_queue or (only_if_queued is not None and not any(q[0] == only_if_queued for q in self._k3_queue)):
            return
        queue, self._k3_queue = self._k3_queue, []
        B = queue[0][4]
        states = []
        for _, am, cand, _, _ in queue:
            hook, am._before_flush = am._before_flush, None
            try:
                states.append(am._device_state(cand.device))
            finally:
                am._before_flush = hook
        if len(queue) > 1 and N.actmax_update_multi_supported(max(q[1].n_latents for q in queue), self.n_collect, B):
            N.actmax_update_multi(states, [q[2] for q in queue], [q[3] for q in queue], B)
        else:
            for (_, am, cand, id_base, _), (vals, ids) in zip(queue, states):
                N.actmax_update(vals, ids, cand, None, id_base, B, N.SL_TIES_ATEN, am._aten_ws(B, cand.device))
        for _, am, _, _, _ in queue:
            am._dev_newer = True

    def _flush_deferred(self):
        """Everything a state read must see: stashed activations of every group, then the queued merges."""
        for gid in range(len(self._groups)):
            self._flush_group(gid)
        self._flush_k3()

    def _flush_group(self, gid: int):
        """Collect whatever the group has stashed, layer by layer (a forward that did not reach every member, or a state read)."""
        group = self._groups[gid]
        stash, group["stash"] = group["stash"], {}
        if stash:
            self._flush_k3()  # a layer's merges run in batch order: queued ones first
        for name, (tensor, version, start, native, module) in stash.items():
            self._check_unmodified(name, tensor, version, start)
            am = self.cache[name]
            hook, am._before_flush = am._before_flush, None
            try:
                am.collect(tensor, native, start, site=(id(module), name))
            finally:
                am._before_flush = hook

    def _finalize(self):
        self._flush_deferred()
        if self._probe is not None:  # hooks removed before a second batch came: no tensor reference outlives the context
            self._probe = {}
        for act_max in self.cache.values():
            act_max.flush()

    def __repr__(self) -> str:
        agg_name = getattr(self.aggregation_fn, "__name__", "custom_function")
        return f"ActMaxCache(layers={list(self.layer_names)}, aggregation_fn='{agg_name}', n_collect={self.n_collect})"

    @property
    def metadata(self) -> dict[str, str]:
        return dict(
            aggregation_fn_name=self.agg_fn_name,
            n_collect=str(self.n_collect),
            layer_names=str(list(self.cache.keys())),
        )

    # ---- on-disk cache: one safetensors file per layer, interchangeable with the reference ----
    # File name  "<aggregation fn>-<n_collect>-<layer>.safetensors"; header keys aggregation_fn_name / n_collect /
    # n_latents / layer_name (all strings); tensors "activations" (bf16) and "sample_ids" (int64)
    # (activation_caching.py:434-465).  A cache either matches this instance completely or it is a miss, and a miss is
    # signalled with FileNotFoundError, which `ActivationComponentVisualizer.run` catches (activation_based.py:331-339).
    def _file_name(self, layer_name: str) -> str:
        return f"{self.agg_fn_name}-{self.n_collect}-{layer_name}.safetensors"

    def _header(self, layer_name: str, act_max: ActMax) -> dict[str, str]:
        return {
            "aggregation_fn_name": self.agg_fn_name,
            "n_collect": str(self.n_collect),
            "n_latents": str(act_max.n_latents),
            "layer_name": layer_name,
        }

    def _usable(self, path: Path) -> str | None:
        """``None`` when ``path`` holds a top-k state collected with this aggregator and ``n_collect``; else why not."""
        if not path.exists():
            return "no such file"
        with safetensors.safe_open(path, framework="pt") as handle:
            header = handle.metadata() or {}
        if header.get("aggregation_fn_name") != self.agg_fn_name:
            return f"aggregated with {header.get('aggregation_fn_name')!r}, this cache uses {self.agg_fn_name!r}"
        try:
            stored_k = int(header.get("n_collect"))
        except (TypeError, ValueError):
            return f"unreadable n_collect {header.get('n_collect')!r}"
        if stored_k != self.n_collect:
            return f"holds the top {stored_k}, this cache collects the top {self.n_collect}"
        return None

    def store(self, directory: Path | str):
        """Write every layer that has seen data; layers without data are skipped with a warning."""
        root = Path(directory)
        root.mkdir(parents=True, exist_ok=True)
        for layer_name, act_max in self.cache.items():
            if act_max.is_setup:
                act_max.store(root / self._file_name(layer_name), metadata=self._header(layer_name, act_max))
            else:
                logger.warning(f"Layer '{layer_name}' has not seen any data; nothing written for it.")
        logger.info(f"Top-k cache written to {root}")

    def load(self, directory: Path | str):
        """Replace every layer's state with the one stored under ``directory`` (activation_caching.py:467-534).

        Raises ``FileNotFoundError`` when the directory, or a usable file for ANY of ``layer_names``, is missing.  The
        files of all layers are validated before the first one is installed, so a miss leaves this cache untouched (the
        reference installs layers as it goes; after a partial hit its caller would collect on top of the layers it did
        load)."""
        root = Path(directory)
        if not root.is_dir():
            raise FileNotFoundError(f"Cache directory not found: {root}")
        paths = {layer_name: root / self._file_name(layer_name) for layer_name in self.layer_names}
        for layer_name, path in paths.items():
            problem = self._usable(path)
            if problem is not None:
                logger.warning(f"Top-k cache miss for layer '{layer_name}' ({path.name}): {problem}")
                raise FileNotFoundError(f"Expected file not found: {path}")
        for layer_name, path in paths.items():
            state = ActMax.load(path)
            state.tie_mode = self.tie_mode
            state.init_value = self.init_value
            self.cache[layer_name] = state
        logger.info(f"Top-k cache: {len(paths)} layer(s) loaded from {root}")
