"""Generate the golden fixtures in tests/golden/ by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Each fixture is data: inputs + the outputs the reference produced for them.
torch 2.10.0 (CPU), scikit-learn 1.7.2, numpy 2.2.6 at generation time.
"""
from __future__ import annotations

import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))

from ref_import import import_reference  # noqa: E402

import_reference()

from helpers import (  # noqa: E402
    FakeVLM,
    TensorPairDataset,
    make_int_conv_model,
    make_int_images,
    tie_free_bf16_matrix,
)
from semanticlens import scores as ref_scores  # noqa: E402
from semanticlens.component_visualization import aggregators as ref_agg  # noqa: E402
from semanticlens.component_visualization.activation_based import ActivationComponentVisualizer  # noqa: E402
from semanticlens.component_visualization.activation_caching import ActMax  # noqa: E402
from semanticlens.lens import Lens, _embed_text_probes, image_probing  # noqa: E402


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    assert t.dtype == torch.bfloat16
    return t.view(torch.int16).numpy().view(np.uint16).copy()


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"wrote {path.name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- 1
def gen_known_answer():
    """tests/component_visualization/test_activation_caching.py:14-30 of the reference."""
    am = ActMax(n_collect=5, n_latents=3)
    a1 = torch.tensor([[0.1, 0.9, 0.3], [0.2, 0.8, 0.4]])
    a2 = torch.tensor([[0.9, 0.1, 0.5], [0.8, 0.2, 0.6]])
    am.update(a1, torch.tensor([0, 1]))
    am.update(a2, torch.tensor([2, 3]))
    # the assertions the reference's test makes (row 0):
    assert torch.allclose(am.activations[0], torch.tensor([0.9, 0.8, 0.2, 0.1, 0.0]).to(torch.bfloat16))
    assert torch.allclose(am.sample_ids[0], torch.tensor([2, 3, 1, 0, -1]))
    save(
        "actmax_known_answer",
        acts1=a1.numpy(),
        ids1=np.array([0, 1]),
        acts2=a2.numpy(),
        ids2=np.array([2, 3]),
        vals=bf16_bits(am.activations),
        ids=am.sample_ids.numpy(),
    )


# --------------------------------------------------------------------------- 2
def run_stream(acts: np.ndarray, k: int, B: int, record_every_batch: bool):
    N, C = acts.shape
    am = ActMax(n_collect=k, n_latents=C)
    hist_v, hist_i = [], []
    for s in range(0, N, B):
        e = min(N, s + B)
        am.update(torch.from_numpy(acts[s:e]), torch.arange(s, e))
        if record_every_batch:
            hist_v.append(bf16_bits(am.activations))
            hist_i.append(am.sample_ids.numpy().copy())
    if record_every_batch:
        return np.stack(hist_v), np.stack(hist_i)
    return bf16_bits(am.activations)[None], am.sample_ids.numpy()[None]


def gen_streams():
    out = {}
    cases = []
    g = np.random.RandomState(11)

    def add(tag, acts, k, B, every):
        v, i = run_stream(acts, k, B, every)
        idx = len(cases)
        cases.append(f"{tag}|k={k}|B={B}|every={int(every)}")
        # inputs stored as raw fp32 (small) — exact
        out[f"acts_{idx}"] = acts
        out[f"vals_{idx}"] = v
        out[f"ids_{idx}"] = i

    # tie-free: indices are well defined, every implementation must agree bit-for-bit
    tf_small = tie_free_bf16_matrix(100, 8, seed=1)
    for k in (0, 1, 5, 20):
        for B in (1, 7, 32):
            add("tiefree", tf_small, k, B, every=True)
    tf_big = tie_free_bf16_matrix(1000, 64, seed=2)
    for k, B in ((20, 64), (100, 64), (20, 256), (100, 37)):
        add("tiefree", tf_big, k, B, every=False)
    # k larger than the dataset: sentinels remain
    add("tiefree", tf_small[:13], 20, 5, every=True)

    # tie-heavy: values on a 1/8 grid (exactly representable), reference tie order is ATen's
    th = (g.randint(-8, 40, size=(300, 16)) / 8.0).astype(np.float32)
    for k, B in ((5, 7), (20, 32), (20, 64), (100, 64)):
        add("tieheavy", th, k, B, every=True)
    # fp32 values that round to bf16 (RNE incl. exact halfway cases)
    rn = g.randn(200, 16).astype(np.float32)
    half = ((np.arange(0x3F80, 0x3F80 + 200, dtype=np.uint32) << 16) | 0x8000).view(np.float32)  # exact ties
    rn[:, 0] = half
    add("rounding", rn, 20, 32, every=True)
    # all negative: nothing displaces the -0.0 sentinels
    add("allneg", -np.abs(g.randn(50, 8)).astype(np.float32) - 0.1, 5, 16, every=True)
    # zeros of both signs + dead channels (ReLU-like)
    z = np.maximum(g.randn(120, 8), 0).astype(np.float32)
    z[:, 1] = 0.0
    z[::3, 2] = -0.0
    add("zeros", z, 10, 32, every=True)
    # NaN and infinities
    sp = g.randn(90, 8).astype(np.float32)
    sp[5, 0] = np.nan
    sp[40, 0] = np.nan
    sp[7, 1] = np.inf
    sp[8, 2] = -np.inf
    sp[60, 3] = np.nan
    add("nan_inf", sp, 5, 30, every=True)
    out["cases"] = np.array(cases)
    save("actmax_streams", **out)


# --------------------------------------------------------------------------- 3
def gen_aggregators():
    g = torch.Generator().manual_seed(21)
    out = {}
    x4a = torch.randn(2, 4, 8, 8, generator=g)
    x4b = torch.randn(3, 5, 7, 7, generator=g)
    x4c = torch.randn(2, 6, 14, 14, generator=g).relu()
    x3a = torch.randn(2, 10, 16, generator=g)
    x3b = torch.randn(3, 197, 24, generator=g)
    x4n = x4a.clone()
    x4n[0, 1, 2, 3] = float("nan")
    for tag, x in (("x4a", x4a), ("x4b", x4b), ("x4c", x4c), ("x4n", x4n)):
        out[tag] = x.numpy()
        out[f"{tag}_conv_mean"] = ref_agg.aggregate_conv_mean(x).numpy()
        out[f"{tag}_conv_max"] = ref_agg.aggregate_conv_max(x).numpy()
    for tag, x in (("x3a", x3a), ("x3b", x3b)):
        out[tag] = x.numpy()
        out[f"{tag}_mean"] = ref_agg.aggregate_transformer_mean(x).numpy()
        out[f"{tag}_absmean"] = ref_agg.aggregate_transformer_absmean(x).numpy()
        out[f"{tag}_max"] = ref_agg.aggregate_transformer_max(x).numpy()
        out[f"{tag}_absmax"] = ref_agg.aggregate_transformer_absmax(x).numpy()
        out[f"{tag}_tok0"] = ref_agg.get_aggregate_transformer_special_token(0)(x).numpy()
        out[f"{tag}_tokm1"] = ref_agg.get_aggregate_transformer_special_token(-1)(x).numpy()
    save("aggregators", **out)


# --------------------------------------------------------------------------- 4
def gen_scores():
    g = torch.Generator().manual_seed(31)
    out = {}
    # similarity_score: the three shape branches (scores.py:119-128)
    x = torch.randn(5, 12, generator=g)
    y = torch.randn(7, 12, generator=g)
    out["sim_x"], out["sim_y"] = x.numpy(), y.numpy()
    out["sim_xyT"] = ref_scores.similarity_score(x, y).numpy()
    y2 = torch.randn(12, 9, generator=g)  # x.shape[1] == y.shape[0] -> no transpose
    out["sim_y2"] = y2.numpy()
    out["sim_xy2"] = ref_scores.similarity_score(x, y2).numpy()
    y3 = torch.randn(5, 12, generator=g)  # equal shapes -> row-wise cosine
    out["sim_y3"] = y3.numpy()
    out["sim_rowwise"] = ref_scores.similarity_score(x, y3).numpy()
    xz = x.clone()
    xz[2] = 0  # zero row: eps clamp
    out["sim_xz"] = xz.numpy()
    out["sim_xzyT"] = ref_scores.similarity_score(xz, y).numpy()
    # larger one, SigLIP-like width
    xl = torch.randn(33, 1152, generator=g)
    yl = torch.randn(70, 1152, generator=g)
    out["sim_xl"], out["sim_yl"] = xl.numpy(), yl.numpy()
    out["sim_xlylT"] = ref_scores.similarity_score(xl, yl).numpy()

    V = torch.randn(10, 20, 128, generator=g)
    out["V"] = V.numpy()
    out["V_clarity"] = ref_scores.clarity_score(V).numpy()
    out["V_redundancy3d"] = ref_scores.redundancy_score(V[:, :15]).numpy()  # 3-D input like tests/test_scores.py:40-51
    cones = V.mean(1)
    out["cones_redundancy"] = ref_scores.redundancy_score(cones).numpy()
    Vz = V[:3].clone()
    Vz[0, 0] = 0
    out["Vz"] = Vz.numpy()
    out["Vz_clarity"] = ref_scores.clarity_score(Vz).numpy()

    # polysemanticity: random rows, two-blob rows, degenerate rows (fallback branch scores.py:173-184)
    P = torch.randn(12, 20, 32, generator=g)
    c1 = torch.randn(12, 1, 32, generator=g) * 3
    c2 = torch.randn(12, 1, 32, generator=g) * 3
    blob = torch.cat([c1 + 0.3 * torch.randn(12, 9, 32, generator=g), c2 + 0.3 * torch.randn(12, 11, 32, generator=g)], 1)
    P[4:8] = blob[4:8]
    P[8] = P[8, :1]  # all samples identical -> one cluster
    P[9, 1:] = P[9, 1:2]  # one outlier + 19 identical -> min count 1 (<2)
    P[10, :10] = P[10, 0:1]
    P[10, 10:] = P[10, 10:11]  # two exact groups of 10
    out["P"] = P.numpy()
    out["P_poly"] = ref_scores.polysemanticity_score(P).numpy()
    P10 = torch.randn(5, 10, 128, generator=g)  # shape used by tests/test_scores.py:54-65
    out["P10"] = P10.numpy()
    out["P10_poly"] = ref_scores.polysemanticity_score(P10).numpy()
    save("scores", **out)


# --------------------------------------------------------------------------- 5
def gen_text_probes():
    out = {}
    fm = FakeVLM()
    queries = ["cat", "dog", "car wheel"]
    templates = ["a photo of a {}", "an image of {}"]
    cases = []
    for qi, qs in enumerate((queries[:1], queries)):
        for ti, ts in enumerate((None, templates[:1], templates)):
            for bs in (None, 2):
                emb = _embed_text_probes(fm, list(qs), ts, bs)
                tag = f"q{len(qs)}_t{0 if ts is None else len(ts)}_bs{bs or 0}"
                cases.append(tag)
                out[tag] = emb.numpy()
    out["cases"] = np.array(cases)
    out["queries"] = np.array(queries)
    out["templates"] = np.array(templates)
    save("text_probes", **out)


# --------------------------------------------------------------------------- 6
def gen_pipeline():
    """End-to-end: Lens.compute_concept_db + text_probing + eval_* on an integer-valued model."""
    from safetensors import safe_open

    out = {}
    model = make_int_conv_model()
    x = make_int_images(40)
    ds = TensorPairDataset(x)
    fm = FakeVLM()
    with tempfile.TemporaryDirectory() as tmp:
        cv = ActivationComponentVisualizer(
            model,
            ds,
            ds,
            layer_names=["0", "2"],
            num_samples=6,
            aggregate_fn=ref_agg.aggregate_conv_max,
            cache_dir=tmp,
        )
        lens = Lens(fm, device="cpu")
        db = lens.compute_concept_db(cv, batch_size=16)
        out["images"] = x.numpy()
        for name in ("0", "2"):
            am = cv.actmax_cache.cache[name]
            out[f"vals_{name}"] = bf16_bits(am.activations)
            out[f"ids_{name}"] = am.sample_ids.numpy()
            out[f"db_{name}"] = db[name].numpy()
        agg_db = {k: v.mean(1) for k, v in db.items()}
        probe = lens.text_probing(["cat", "dog"], agg_db, templates=["a photo of a {}"])
        for name in ("0", "2"):
            out[f"probe_{name}"] = probe[name].numpy()
            out[f"clarity_{name}"] = lens.eval_clarity(db)[name].numpy()
            out[f"redundancy_{name}"] = np.asarray(lens.eval_redundancy(agg_db)[name].numpy())
        # on-disk layout of the caches: file names + metadata
        files = sorted(str(p.relative_to(tmp)) for p in Path(tmp).rglob("*.safetensors"))
        out["cache_files"] = np.array(files)
        metas = []
        for f in files:
            with safe_open(os.path.join(tmp, f), framework="pt") as fh:
                metas.append(repr(sorted((fh.metadata() or {}).items())) + "|" + repr(sorted(fh.keys())))
        out["cache_meta"] = np.array(metas)
    # embed-all-then-gather check data
    out["embeds"] = fm.encode_image(fm.preprocess([ds[i][0] for i in range(len(ds))])).numpy()
    save("pipeline", **out)


# --------------------------------------------------------------------------- 7
def gen_image_probes():
    """lens.py:124-162 — one image (no averaging) and several images (mean of their embeddings) against a
    tensor DB and a dict DB; FakeVLM's integer projection makes the query embedding exact."""
    out = {}
    fm = FakeVLM()
    imgs = make_int_images(3, seed=9)
    g = torch.Generator().manual_seed(41)
    db = torch.randn(7, fm.dim, generator=g)
    db_dict = {"a": torch.randn(5, fm.dim, generator=g), "b": torch.randn(1, fm.dim, generator=g)}
    out["images"] = imgs.numpy()
    out["db"] = db.numpy()
    out["db_a"], out["db_b"] = db_dict["a"].numpy(), db_dict["b"].numpy()
    for tag, query in (("one", imgs[0]), ("one_list", [imgs[1]]), ("three", [imgs[0], imgs[1], imgs[2]])):
        out[f"{tag}_tensor"] = image_probing(fm, query, db).numpy()
        res = image_probing(fm, query, db_dict)
        out[f"{tag}_a"], out[f"{tag}_b"] = res["a"].numpy(), res["b"].numpy()
    emb = fm.encode_image(fm.preprocess([imgs[0], imgs[1], imgs[2]]))
    out["three_query_embed"] = emb.mean(0)[None].numpy()
    save("image_probes", **out)


def gen_scores_k():
    """polysemanticity_score(n_clusters=k) of the reference for k != 2 (scores.py:132,167): random rows, three / four
    well-separated blobs, rows with duplicated points (empty clusters, the < 2 samples fallback)."""
    import warnings

    g = torch.Generator().manual_seed(11)
    P = torch.randn(14, 24, 32, generator=g)
    cents = torch.randn(14, 4, 1, 32, generator=g) * 3
    for i in range(4, 8):  # three blobs of 8
        P[i] = torch.cat([cents[i, j] + 0.3 * torch.randn(8, 32, generator=g) for j in range(3)], 0)
    for i in range(8, 11):  # four blobs of 6
        P[i] = torch.cat([cents[i, j] + 0.3 * torch.randn(6, 32, generator=g) for j in range(4)], 0)
    P[11, 4:] = P[11, 4:5]  # 4 distinct points + 20 copies
    P[12] = P[12, :1]  # all identical
    P[13, :12] = P[13, 0:1]
    P[13, 12:] = P[13, 12:13]  # two exact groups
    out = {"P": P.numpy()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # sklearn's ConvergenceWarning on the degenerate rows
        for k in (3, 4, 6):
            out[f"poly_k{k}"] = ref_scores.polysemanticity_score(P, n_clusters=k).numpy()
    save("scores_k", **out)


def gen_aggregators_half():
    """The reference's aggregators on fp16 / bf16 activations (half-precision models): what dtype comes back and how the
    means are rounded (torch reduces in fp32 and rounds once to the tensor's dtype)."""
    g = torch.Generator().manual_seed(23)
    out = {}
    x4 = {"h4a": torch.randn(3, 5, 7, 7, generator=g), "h4b": torch.randn(2, 8, 14, 14, generator=g) * 4,
          "h4c": torch.randn(2, 6, 28, 28, generator=g).relu()}
    x3 = {"h3a": torch.randn(2, 50, 24, generator=g), "h3b": torch.randn(3, 197, 16, generator=g) * 3}
    for dname, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        for tag, x in x4.items():
            xh = x.to(dt)
            out[f"{tag}_{dname}"] = xh.float().numpy()  # the rounded input, exactly representable
            for fn in ("aggregate_conv_mean", "aggregate_conv_max"):
                r = getattr(ref_agg, fn)(xh)
                assert r.dtype == dt
                out[f"{tag}_{dname}_{fn}"] = r.float().numpy()
        for tag, x in x3.items():
            xh = x.to(dt)
            out[f"{tag}_{dname}"] = xh.float().numpy()
            for fn in ("aggregate_transformer_mean", "aggregate_transformer_absmean", "aggregate_transformer_max", "aggregate_transformer_absmax"):
                r = getattr(ref_agg, fn)(xh)
                assert r.dtype == dt
                out[f"{tag}_{dname}_{fn}"] = r.float().numpy()
    save("aggregators_half", **out)


GENERATORS = {
    "aggregators_half": gen_aggregators_half,
    "scores_k": gen_scores_k,
    "known_answer": gen_known_answer,
    "streams": gen_streams,
    "aggregators": gen_aggregators,
    "scores": gen_scores,
    "text_probes": gen_text_probes,
    "pipeline": gen_pipeline,
    "image_probes": gen_image_probes,
}

if __name__ == "__main__":
    # `python make_golden.py` regenerates everything; `python make_golden.py image_probes ...` only those named
    for name in sys.argv[1:] or list(GENERATORS):
        torch.manual_seed(0)
        GENERATORS[name]()
