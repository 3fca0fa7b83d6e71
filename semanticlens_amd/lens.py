"""Lens — orchestration of concept-DB build, probing and evaluation.

Mirror of ``semanticlens/lens.py`` (reference v0.2.1): same module-level functions and ``Lens``
methods, same cache-file naming.  Text/image probing runs the template-difference mean (K10)
and the cosine GEMM (K6) on the device; query embeddings are not bounced through the host
between the text tower and the GEMM.
"""
from __future__ import annotations

import os

import logging

import torch
from safetensors.torch import load_file, save_file
from tqdm.auto import tqdm

from semanticlens_amd import _native as N
from semanticlens_amd.component_visualization.base import AbstractComponentVisualizer
from semanticlens_amd.foundation_models.base import AbstractVLM
from semanticlens_amd.scores import clarity_score, polysemanticity_score, redundancy_score, similarity_score
from semanticlens_amd.utils.helper import get_fallback_name

logger = logging.getLogger(__name__)


def compute_concept_db(cv: AbstractComponentVisualizer, fm: AbstractVLM):
    """Stateless concept-DB build: delegates to ``cv._compute_concept_db(fm)`` (lens.py:27-56)."""
    return cv._compute_concept_db(fm)


def text_probing(fm, query, aggregated_concept_db, templates=None, batch_size=None):
    """Cosine similarity of text queries against ``(n_components, D)`` concept embeddings (lens.py:59-121)."""
    queries = query if isinstance(query, list) else [query]
    query_embeds = _embed_text_probes(fm, queries, templates, batch_size)
    assert query_embeds.ndim == 2
    assert query_embeds.shape[0] == len(queries)
    return _probe(query_embeds, aggregated_concept_db)


def image_probing(fm, query, aggregated_concept_db):
    """Cosine similarity of an image query (several images are averaged) against the DB (lens.py:124-162)."""
    return _probe(_embed_image_probe(fm, query), aggregated_concept_db)


@torch.no_grad()
def _embed_image_probe(fm, query) -> torch.Tensor:
    """``(1, D)`` probe vector of one image or of a list of images (lens.py:158-160).

    Several images are averaged on the HOST, as the reference does (it moves the ``(n_images, D)`` embeddings to the
    CPU first): torch's device ``mean`` multiplies by a reciprocal where the CPU one divides, which differs in the last
    bit; the tensor is a few KB."""
    embeds = fm.encode_image(fm.preprocess(query).to(fm.device))
    if embeds.shape[0] > 1:
        return embeds.cpu().mean(0)[None]
    return embeds


def _encode_texts(fm, texts: list[str], batch_size: int | None = None, progress: bool = False):
    """``fm.encode_text(fm.tokenize(chunk))`` over ``texts`` in chunks of ``batch_size`` (lens.py:176-191).

    With several chunks the tokenizer (host Python: 6-9 ms per 1 024 prompts for the bench's stand-in, more for a BPE) runs one
    chunk AHEAD on a helper thread while the device encodes the current one — the text tower waits on a readback per batch, which
    releases the GIL — so a 10 000-prompt probe costs max(tokenise, encode) instead of their sum.  Same calls, same order, same
    results; ``SL_TEXT_PREFETCH=0`` restores the serial loop."""
    batch_size = batch_size or len(texts)
    starts = list(range(0, len(texts), batch_size))
    chunks = []
    bar = tqdm(starts, desc="text embedding ...", leave=False, disable=not progress or batch_size >= len(texts))
    if len(starts) > 1 and os.environ.get("SL_TEXT_PREFETCH", "1") != "0":
        from concurrent.futures import ThreadPoolExecutor

        # A new thread's current HIP device is 0 and its current stream the default one.  A wrapper's `tokenize` may end in
        # `.to(self.device)`: with an index-less device ("cuda") that copy would land on GPU 0 on every rank, issued on a stream
        # the encoder does not run on.  The helper therefore tokenises under the CALLER's device and stream; `.to(fm.device)` on
        # the main thread is then a no-op for tensors already there.
        dev = torch.device(fm.device)
        if dev.type == "cuda":
            cur_dev = torch.cuda.current_device() if dev.index is None else dev.index
            cur_stream = torch.cuda.current_stream(cur_dev)

            def tokenize(chunk):
                with torch.cuda.device(cur_dev), torch.cuda.stream(cur_stream):
                    return fm.tokenize(chunk)
        else:
            tokenize = fm.tokenize
        with ThreadPoolExecutor(max_workers=1) as pool:
            nxt = pool.submit(tokenize, texts[starts[0] : starts[0] + batch_size])
            for i, _ in enumerate(bar):
                tokens = nxt.result()
                if i + 1 < len(starts):
                    nxt = pool.submit(tokenize, texts[starts[i + 1] : starts[i + 1] + batch_size])
                chunks.append(fm.encode_text(tokens.to(fm.device)))
    else:
        for start in bar:
            chunks.append(fm.encode_text(fm.tokenize(texts[start : start + batch_size]).to(fm.device)))
    return chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=0)


@torch.no_grad()
def _embed_text_probes(fm, query: list[str], templates: list[str] | None, batch_size: int | None, encode=None):
    """Tokenise + encode the (templated) queries; with templates, subtract the empty-template
    embedding and average over templates (lens.py:165-203).

    The reference builds the templated list template-major (``for t in templates for q in query``,
    :174) but regroups it query-major (``"(q t) d -> q t d"``, :197).  That grouping is kept
    (SURVEY.md finding 4) so probing scores equal the reference's.

    ``encode(texts, batch_size)`` replaces the local text tower (``distributed.text_probing_sharded`` shards the
    prompt list across ranks there); each prompt's embedding does not depend on how the list is split.
    """
    if templates:
        query_templated = [t.format(q) for t in templates for q in query]
        empty_templates = [t.format("") for t in templates]
        if encode is None:
            templated = _encode_texts(fm, query_templated, batch_size, progress=True)
        else:
            templated = encode(query_templated, batch_size)
        empty = fm.encode_text(fm.tokenize(empty_templates).to(fm.device))
        return N.template_mean(templated, empty, len(query))
    if encode is not None:
        return encode(query, None)
    return fm.encode_text(fm.tokenize(query).to(fm.device))


@torch.no_grad()
def _probe(query: torch.Tensor, aggregated_concept_db):
    if isinstance(aggregated_concept_db, torch.Tensor):
        return similarity_score(query.to(aggregated_concept_db.device), aggregated_concept_db)
    keys = list(aggregated_concept_db)
    values = [aggregated_concept_db[k] for k in keys]
    # all layers in one native call (query normalised + split once) when none of them hits a shape quirk
    if values and all(isinstance(v, torch.Tensor) for v in values) and len({v.device for v in values}) == 1:
        outs = N.similarity_multi(query, values)
        if outs is not None:
            return {k: o.to(v.device) for k, o, v in zip(keys, outs, values)}
    return {key: similarity_score(query.to(value.device), value) for key, value in aggregated_concept_db.items()}


class Lens:
    """Holds the foundation model and wraps the workflow (reference: lens.py:217-480)."""

    def __init__(self, fm, device=None):
        self.fm = fm
        self.device = device or self.fm.device
        self.fm.to(self.device)
        if not hasattr(self.fm, "name"):
            self.fm.name = get_fallback_name(self.fm)
            logger.debug(f"Assigned fallback name to foundation model: {self.fm.name}")

    def compute_concept_db(self, cv: AbstractComponentVisualizer, **kwargs) -> dict[str, torch.Tensor]:
        """Build the concept DB through ``cv``, or load it from ``cv``'s cache directory.

        Cache file: ``<storage_dir>/concept_database/<fm.name>/concept_db-<metadata values except
        dataset,model joined by '-'>.safetensors`` (lens.py:308-316).
        """
        if not cv.caching:
            return cv._compute_concept_db(self.fm, **kwargs)
        path = self._concept_db_path(cv)
        if path.exists():
            logger.debug(f"concept DB read from {path}")
            return load_file(filename=path)
        concept_db = cv._compute_concept_db(self.fm, **kwargs)
        save_file(tensors=concept_db, filename=path)
        logger.debug(f"concept DB written to {path}")
        return concept_db

    def _concept_db_path(self, cv):
        """The cache file of ``cv``'s concept DB under this foundation model; creates its directory."""
        folder = cv.storage_dir / "concept_database" / self.fm.name
        folder.mkdir(parents=True, exist_ok=True)
        tags = [value for key, value in cv.metadata.items() if key not in ("dataset", "model")]
        return folder / ("concept_db-" + "-".join(tags) + ".safetensors")

    def text_probing(self, query, aggregated_concept_db, templates=None, batch_size=None):
        return text_probing(self.fm, query, aggregated_concept_db, templates, batch_size)

    def image_probing(self, query, aggregated_concept_db):
        return image_probing(self.fm, query, aggregated_concept_db)

    @staticmethod
    def _per_layer(fn, db):
        if isinstance(db, torch.Tensor):
            return fn(db)
        return {key: fn(value) for key, value in db.items()}

    def eval_clarity(self, concept_db):
        """``clarity_score`` of a ``(C, n, D)`` tensor or of each layer of a dict (lens.py:391-419)."""
        if isinstance(concept_db, dict) and len(concept_db) > 1:
            # the per-layer loop of lens.py:391-419 as one launch over all layers (K7 reads C*n*D*4 bytes once; a layer alone is
            # a few MB — launch-latency-sized); same values as layer by layer
            keys = list(concept_db)
            if all(isinstance(concept_db[k_], torch.Tensor) for k_ in keys):
                outs = N.clarity_multi([concept_db[k_] for k_ in keys])
                if outs is not None:
                    return {k_: o.to(concept_db[k_].device) for k_, o in zip(keys, outs)}
        return self._per_layer(clarity_score, concept_db)

    def eval_redundancy(self, aggregated_concept_db):
        """``redundancy_score`` of a ``(C, D)`` tensor or dict of tensors (lens.py:421-449)."""
        return self._per_layer(redundancy_score, aggregated_concept_db)

    def eval_polysemanticity(self, concept_db):
        """``polysemanticity_score`` of a ``(C, n, D)`` tensor or dict of tensors (lens.py:451-480)."""
        return self._per_layer(polysemanticity_score, concept_db)
