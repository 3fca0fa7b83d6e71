"""CPU-only tests (-m "not gpu"): C-ABI surface, host logic mirrored from the reference's own unit tests
(reference tests/… cited per test), cache formats, loud failure without a HIP device."""
import re
import subprocess
import sys
from pathlib import Path
from unittest import mock

import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.utils.data import TensorDataset

from semanticlens_amd import Lens
from semanticlens_amd import _native as N
from semanticlens_amd.component_visualization import ActivationComponentVisualizer
from semanticlens_amd.component_visualization import aggregators as agg
from semanticlens_amd.component_visualization.activation_based import MissingNameWarning
from semanticlens_amd.component_visualization.activation_caching import ActMax, ActMaxCache
from semanticlens_amd.utils import get_fallback_name

ROOT = Path(__file__).resolve().parent.parent
NO_GPU = not torch.cuda.is_available()


# ------------------------------------------------------------------------------------------ C ABI
def test_library_loads_and_exports_every_declared_symbol():
    header = (ROOT / "include" / "semanticlens_amd.h").read_text()
    declared = set(re.findall(r"\b(sl_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    handle = N.lib()  # resolves every name in N.SIGNATURES or raises AttributeError
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    nm = subprocess.run(["nm", "-D", str(N.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sl_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    assert handle.sl_abi_version() == 1
    assert "gfx950" in subprocess.run(["strings", str(N.LIB_PATH)], capture_output=True, text=True).stdout


def test_argument_errors_are_reported_without_a_device():
    lib = N.lib()
    assert lib.sl_actmax_merge(None, None, 4, 5, None, 0, None, None, 1, None) == -1
    assert lib.sl_last_error() == b"sl_actmax_merge: null state"
    assert lib.sl_actmax_merge(1, 1, 4, 5000, None, 0, None, None, 1, None) == -1  # validated before any launch
    assert b"exceeds the supported maximum" in lib.sl_last_error()
    assert lib.sl_similarity(None, 3, 4, None, 5, 6, None, None, 0, None) == -1
    assert lib.sl_last_error() == b"x and y must have the same shape"  # the reference's ValueError text
    assert lib.sl_reduce_conv(None, 7, 1, 1, 1, 1, 1, 1, 0, None, None, None) == -1
    # the communicator entry points validate before they touch RCCL or a device
    assert lib.sl_comm_allgather(None, None, None, 16, None) == -1 and lib.sl_last_error() == b"sl_comm_allgather: null communicator"
    assert lib.sl_comm_allreduce(None, None, 4, 0, 0, None) == -1
    assert lib.sl_actmax_allgather_merge(None, 1, None, None, None, 5, None, 0, None) == -1
    assert lib.sl_comm_init_from_unique_id(None, 2, 0, None) == -1
    assert lib.sl_comm_destroy(None) == 0  # destroying nothing is fine
    import ctypes

    C = (ctypes.c_int64 * 3)(512, 1024, 2048)
    per_rank = lib.sl_actmax_packed_bytes(3, C, 20)
    assert per_rank == 3584 * 20 * 10 and lib.sl_actmax_allgather_merge_ws_bytes(3, C, 20, 8) == 9 * per_rank  # 717 KB per rank
    assert lib.sl_actmax_packed_bytes(1, (ctypes.c_int64 * 1)(3), 5) == 160  # 150 bytes rounded up to 16
    assert lib.sl_actmax_merge_packed(1, None, None, None, 5, None, 2, -1, None) == -1


@pytest.mark.skipif(not NO_GPU, reason="checks the no-device behaviour")
def test_no_cpu_fallback_compute_raises_without_device():
    from semanticlens_amd import scores

    with pytest.raises(N.NativeLibraryError, match="no CPU fallback"):
        scores.similarity_score(torch.randn(3, 4), torch.randn(5, 4))
    with pytest.raises(N.NativeLibraryError):
        agg.aggregate_conv_max(torch.randn(2, 3, 4, 4))
    am = ActMax(n_collect=2, n_latents=3)
    with pytest.raises(N.NativeLibraryError):
        am.update(torch.randn(4, 3), torch.arange(4))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setenv("SEMANTICLENS_AMD_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(N.NativeLibraryError, match="no CPU fallback"):
        N.lib()


# ------------------------------------------------------------------------------------- aggregators
def test_aggregator_names_and_argument_checks():
    # names are part of the cache file names (reference aggregators.py:27,32)
    for name in ("aggregate_conv_mean", "aggregate_conv_max", "aggregate_transformer_mean", "aggregate_transformer_absmean",
                 "aggregate_transformer_max", "aggregate_transformer_absmax"):
        assert getattr(agg, name).__name__ == name
    assert agg.get_aggregate_transformer_special_token(0).__name__ == "aggregate_transformer_special_token"
    # reference tests/component_visualization/test_aggregators.py:28-31, 51-54
    with pytest.raises(ValueError, match="Input tensor should be 4D"):
        agg.aggregate_conv_mean(torch.randn(2, 4, 8))
    with pytest.raises(ValueError, match="Input tensor should be 3D"):
        agg.aggregate_transformer_max(torch.randn(2, 10, 16, 1))
    with pytest.raises(AttributeError):  # tuple outputs fail on `.ndim`, as in the reference
        agg.aggregate_conv_max((torch.randn(2, 3, 4, 4),))


# ------------------------------------------------------------------------------------------ ActMax
def test_actmax_initial_state_and_store_load_roundtrip(tmp_path):
    am = ActMax(n_collect=5, n_latents=3)
    assert am.is_setup and am.activations.dtype == torch.bfloat16 and am.sample_ids.dtype == torch.int64
    assert torch.equal(am.sample_ids, -torch.ones(3, 5, dtype=torch.int64))
    assert np.all(am.activations.view(torch.int16).numpy().view(np.uint16) == 0x8000)  # -0.0
    assert am.alive_latents.numel() == 0
    lazy = ActMax(n_collect=5)
    assert not lazy.is_setup and lazy.alive_latents.numel() == 0
    # store / load (reference tests/component_visualization/test_activation_caching.py:32-48)
    am.activations = torch.rand(3, 5).to(torch.bfloat16)
    am.sample_ids = torch.arange(15).reshape(3, 5)
    path = tmp_path / "actmax.safetensors"
    am.store(path, metadata={"n_collect": "5", "n_latents": "3"})
    back = ActMax.load(path)
    assert back.n_collect == 5 and back.n_latents == 3
    assert torch.equal(back.activations, am.activations) and torch.equal(back.sample_ids, am.sample_ids)
    assert set(back.alive_latents.tolist()) <= {0, 1, 2}


def test_actmax_cache_naming_metadata_and_validation(tmp_path):
    with pytest.raises(ValueError, match="must be a defined function, not a lambda"):
        ActMaxCache(["a"], lambda t: t, 3)
    with pytest.raises(ValueError, match="tie_mode"):
        ActMaxCache(["a"], agg.aggregate_conv_max, 3, tie_mode="bogus")
    cache = ActMaxCache(["layer4", "fc"], agg.aggregate_conv_max, 7)
    assert cache.metadata == {"aggregation_fn_name": "aggregate_conv_max", "n_collect": "7", "layer_names": "['layer4', 'fc']"}
    assert repr(cache) == "ActMaxCache(layers=['layer4', 'fc'], aggregation_fn='aggregate_conv_max', n_collect=7)"
    for name in ("layer4", "fc"):
        cache.cache[name] = ActMax(7, 4)
    cache.store(tmp_path / "c")
    assert sorted(p.name for p in (tmp_path / "c").iterdir()) == [
        "aggregate_conv_max-7-fc.safetensors", "aggregate_conv_max-7-layer4.safetensors"]
    again = ActMaxCache(["layer4", "fc"], agg.aggregate_conv_max, 7)
    again.load(tmp_path / "c")
    assert again["fc"].n_latents == 4
    # any mismatch is a cache miss = FileNotFoundError (activation_caching.py:505-525)
    with pytest.raises(FileNotFoundError):
        ActMaxCache(["layer4"], agg.aggregate_conv_mean, 7).load(tmp_path / "c")
    with pytest.raises(FileNotFoundError):
        ActMaxCache(["layer4"], agg.aggregate_conv_max, 8).load(tmp_path / "c")
    with pytest.raises(FileNotFoundError):
        ActMaxCache(["layer4"], agg.aggregate_conv_max, 7).load(tmp_path / "missing")


def test_tie_mode_default_comes_from_environment(monkeypatch):
    monkeypatch.delenv("SEMANTICLENS_AMD_TIES", raising=False)
    assert ActMax(3).tie_mode == "aten"
    monkeypatch.setenv("SEMANTICLENS_AMD_TIES", "total")
    assert ActMax(3).tie_mode == "total"
    monkeypatch.setenv("SEMANTICLENS_AMD_TIES", "nope")
    with pytest.raises(ValueError):
        ActMax(3)


# ------------------------------------------------------------------- ActivationComponentVisualizer
@pytest.fixture
def mock_model():
    model = nn.Sequential(nn.Conv2d(3, 8, 3), nn.ReLU(), nn.Conv2d(8, 16, 3))
    model.name = "mock_model"
    return model


@pytest.fixture
def mock_dataset():
    ds = TensorDataset(torch.randn(4, 3, 32, 32), torch.randn(4, 3, 32, 32))
    ds.name = "mock_dataset"
    return ds


def test_visualizer_initialisation_and_errors(mock_model, mock_dataset, tmp_path):
    # reference tests/component_visualization/test_activation_based.py:26-67
    cv = ActivationComponentVisualizer(mock_model, mock_dataset, mock_dataset, ["0"], num_samples=10, cache_dir=None)
    assert cv.model is mock_model and cv.layer_names == ["0"] and not cv.caching
    assert cv.actmax_cache.agg_fn_name == "aggregate_conv_mean"  # default aggregator
    assert cv.metadata == {"aggregation_fn_name": "aggregate_conv_mean", "n_collect": "10", "layer_names": "['0']",
                           "dataset": "mock_dataset", "model": "mock_model"}
    with pytest.raises(ValueError, match="Layer 'bad_layer' not found in model"):
        ActivationComponentVisualizer(mock_model, mock_dataset, mock_dataset, ["bad_layer"], num_samples=10)
    with pytest.raises(ValueError, match="should have the same length"):
        ActivationComponentVisualizer(mock_model, mock_dataset, TensorDataset(torch.randn(3, 1)), ["0"], num_samples=1)
    with pytest.raises(ValueError, match="not found in model layers"):
        cv.get_max_reference("2")
    del mock_model.name
    with pytest.warns(MissingNameWarning, match="Model does not have a name attribute"):
        cv2 = ActivationComponentVisualizer(mock_model, mock_dataset, mock_dataset, ["0"], 10, cache_dir=str(tmp_path))
    assert mock_model.name == get_fallback_name(mock_model) and mock_model.name.startswith("Sequential-")
    assert cv2.storage_dir == tmp_path / "ActivationComponentVisualizer" / "mock_dataset" / mock_model.name


def test_run_uses_cache_when_available(mock_model, mock_dataset, tmp_path):
    # reference test_activation_based.py:70-93: load called twice (ctor + run), _run never
    with mock.patch.object(ActMaxCache, "load", return_value={}) as load, mock.patch.object(
        ActivationComponentVisualizer, "_run"
    ) as run:
        cv = ActivationComponentVisualizer(mock_model, mock_dataset, mock_dataset, ["0"], 10, cache_dir=str(tmp_path))
        cv.run()
    assert load.call_count == 2
    run.assert_not_called()


def test_run_computes_on_cache_miss(mock_model, mock_dataset, tmp_path):
    # reference test_activation_based.py:96-123 (the compute itself is patched: no device here)
    with mock.patch.object(ActMaxCache, "load", side_effect=FileNotFoundError), mock.patch.object(
        ActivationComponentVisualizer, "_run", return_value="computed"
    ) as run:
        cv = ActivationComponentVisualizer(mock_model, mock_dataset, mock_dataset, ["0"], 10, cache_dir=str(tmp_path))
        assert cv.run(batch_size=2) == "computed"
    run.assert_called_once_with(batch_size=2, num_workers=0)


def test_empty_layer_list_runs_without_a_device(mock_model, mock_dataset):
    # reference test_activation_based.py:126-139
    cv = ActivationComponentVisualizer(mock_model, mock_dataset, mock_dataset, [], num_samples=10)
    assert cv.run() == {}


# -------------------------------------------------------------------------------------------- Lens
@pytest.fixture
def mock_fm():
    fm = mock.MagicMock()
    fm.device = "cpu"
    fm.to.return_value = fm
    return fm


@pytest.fixture
def mock_cv(tmp_path):
    cv = mock.MagicMock()
    cv.caching = True
    cv.storage_dir = tmp_path
    cv._compute_concept_db.return_value = {"layer1": torch.randn(10, 5, 128)}
    cv.metadata = {"aggregation_fn_name": "aggregate_conv_max", "n_collect": "5", "layer_names": "['layer1']",
                   "dataset": "d", "model": "m"}
    return cv


def test_lens_moves_fm_and_names_it(mock_fm):
    # reference tests/test_lens.py:35-42
    lens = Lens(fm=mock_fm, device="cpu")
    assert lens.fm is mock_fm
    mock_fm.to.assert_called_with("cpu")

    class Nameless:
        device = "cpu"

        def to(self, d):
            return self

    fm = Nameless()
    Lens(fm)
    assert fm.name == get_fallback_name(fm)


def test_lens_concept_db_cache_miss_then_hit(mock_fm, mock_cv, tmp_path):
    # reference tests/test_lens.py:45-81, with the real safetensors round trip
    mock_fm.name = "fm-x"
    lens = Lens(fm=mock_fm)
    db = lens.compute_concept_db(mock_cv, batch_size=8)
    mock_cv._compute_concept_db.assert_called_once_with(mock_fm, batch_size=8)
    f = tmp_path / "concept_database" / "fm-x" / "concept_db-aggregate_conv_max-5-['layer1'].safetensors"
    assert f.exists()  # lens.py:308-316 naming
    db2 = lens.compute_concept_db(mock_cv)
    mock_cv._compute_concept_db.assert_called_once()  # served from the cache
    assert torch.equal(db2["layer1"], db["layer1"])
    mock_cv.caching = False
    lens.compute_concept_db(mock_cv)
    assert mock_cv._compute_concept_db.call_count == 2


def test_text_probing_host_flow_with_patched_kernels(mock_fm):
    # reference tests/test_lens.py:84-97: one encode_text call, (1, n_components) per layer
    mock_fm.encode_text.return_value = torch.randn(1, 128)
    lens = Lens(fm=mock_fm)
    with mock.patch("semanticlens_amd.lens.similarity_score", side_effect=lambda q, d: torch.zeros(q.shape[0], d.shape[0])), \
            mock.patch("semanticlens_amd.lens.N.similarity_multi", return_value=None):  # no device here: per-layer path
        res = lens.text_probing("a test query", {"layer1": torch.randn(10, 128)})
    mock_fm.encode_text.assert_called_once()
    assert res["layer1"].shape == (1, 10)


def test_text_probe_template_order_matches_reference(golden):
    """lens.py:174 builds template-major; the regrouping quirk happens in K10.  Check the strings/batching."""
    from helpers import FakeVLM
    from semanticlens_amd import lens as L

    fm = FakeVLM()
    seen = []
    orig = fm.tokenize
    fm.tokenize = lambda txt: (seen.append(list(txt)), orig(txt))[1]
    with mock.patch.object(L.N, "template_mean", side_effect=lambda E, E0, Q: torch.zeros(Q, E.shape[1])) as tm:
        out = L._embed_text_probes(fm, ["cat", "dog"], ["a photo of a {}", "an image of {}"], 3)
    assert seen == [["a photo of a cat", "a photo of a dog", "an image of cat"], ["an image of dog"],
                    ["a photo of a ", "an image of "]]
    E, E0, Q = tm.call_args[0]
    assert E.shape == (4, 16) and E0.shape == (2, 16) and Q == 2 and out.shape == (2, 16)


# ---------------------------------------------------------------------------------------------- K9
def test_kmeans_draws_match_sklearn_seeding():
    """The host-side draws equal what scikit-learn's KMeans consumes: replay them against sklearn's own
    k-means++ (first centre = our draw; the candidate thresholds = our uniforms x potential)."""
    from sklearn.cluster import _kmeans

    from semanticlens_amd.scores import kmeans_draws

    n, n_init, seed = 20, 10, 123
    X = np.random.RandomState(0).randn(n, 8)  # float64, what the reference's torch-tensor input becomes inside sklearn
    norms = (X * X).sum(1)
    sw = np.ones(n, dtype=np.float64)
    for k in (2, 3, 8):
        trials = 2 + int(np.log(k))
        first, rand = kmeans_draws(n, n_init, seed, k)
        assert rand.shape == (n_init, k - 1, trials)
        rs = np.random.RandomState(seed)
        for i in range(n_init):
            state_before = rs.get_state()
            centers, idx = _kmeans._kmeans_plusplus(X, k, norms, sw, rs)
            assert idx[0] == first[i]
            replay = np.random.RandomState()
            replay.set_state(state_before)
            replay.choice(n, p=sw / sw.sum())
            for c in range(k - 1):
                assert np.array_equal(replay.uniform(size=trials), rand[i, c])
            assert replay.get_state()[1].tolist() == rs.get_state()[1].tolist() and replay.get_state()[2] == rs.get_state()[2]


# ------------------------------------------------------------------------------------ distributed
def test_shard_ranges_cover_dataset():
    from semanticlens_amd.distributed import shard_range

    for n, w in ((10, 3), (1280000, 8), (5, 8), (0, 2), (50176, 1)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(0 <= s <= e for s, e in spans)


def test_pack_unpack_states_roundtrip():
    from semanticlens_amd.distributed import pack_states, unpack_states

    states = [(torch.randn(4, 3).to(torch.bfloat16), torch.randint(-1, 99, (4, 3))),
              (torch.randn(2, 5).to(torch.bfloat16), torch.randint(-1, 99, (2, 5)))]
    buf = pack_states(states)
    assert buf.dtype == torch.uint8 and buf.numel() == 224  # (12 + 10) * 10 = 220, padded to 16
    back = unpack_states(buf, [(4, 3), (2, 5)])
    for (v, i), (v2, i2) in zip(states, back):
        assert torch.equal(v, v2) and torch.equal(i, i2)


def test_aten_order_restatement_matches_libstdcxx(tmp_path):
    """csrc/aten_topk_order.hpp (compiled for the host) vs std::partial_sort / nth_element / sort."""
    exe = tmp_path / "aten_order_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-include", "cstring", "-o", str(exe),
                    str(ROOT / "tests" / "native" / "aten_order_check.cpp")], check=True)
    res = subprocess.run([str(exe), "20000"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "mismatches=0" in res.stdout


# ------------------------------------------------------------- cache files interchange with the reference
@pytest.mark.skipif(not Path("/root/reference/semanticlens").exists(), reason="reference only exists in the build container")
def test_cache_files_interchange_with_reference(tmp_path):
    """a cache written by either package is accepted by the other (names, metadata, dtypes)."""
    sys.path.insert(0, str(ROOT / "tests" / "golden"))
    from ref_import import import_reference

    import_reference()
    from semanticlens.component_visualization import aggregators as ref_agg
    from semanticlens.component_visualization.activation_caching import ActMaxCache as RefCache

    ours = ActMaxCache(["layer4", "fc"], agg.aggregate_conv_max, 5)
    for i, name in enumerate(("layer4", "fc")):
        am = ActMax(5, 6)
        am.activations = (torch.rand(6, 5) + i).to(torch.bfloat16)
        am.sample_ids = torch.randint(-1, 1000, (6, 5))
        ours.cache[name] = am
    ours.store(tmp_path / "ours")
    ref = RefCache(["layer4", "fc"], ref_agg.aggregate_conv_max, 5)
    ref.load(tmp_path / "ours")
    for name in ("layer4", "fc"):
        assert torch.equal(ref.cache[name].activations, ours.cache[name].activations)
        assert torch.equal(ref.cache[name].sample_ids, ours.cache[name].sample_ids)
    ref.store(tmp_path / "theirs")
    back = ActMaxCache(["layer4", "fc"], agg.aggregate_conv_max, 5)
    back.load(tmp_path / "theirs")
    assert torch.equal(back["fc"].activations, ours["fc"].activations)
    assert sorted(p.name for p in (tmp_path / "ours").iterdir()) == sorted(p.name for p in (tmp_path / "theirs").iterdir())
    import json
    import struct

    def parse(path):  # safetensors: u64 header length, JSON header, payload
        raw = path.read_bytes()
        (n,) = struct.unpack("<Q", raw[:8])
        return json.loads(raw[8 : 8 + n]), raw[8 + n :]

    for name in sorted(p.name for p in (tmp_path / "ours").iterdir()):
        ho, po = parse(tmp_path / "ours" / name)
        ht, pt = parse(tmp_path / "theirs" / name)
        assert ho == ht and po == pt  # same header (metadata key order is unordered in safetensors) and same bytes


# ---- K12 host side: the launch plan (pure host function of the library) and batch packing -----------------------
def test_preprocess_plan_matches_torchvision_size_rules():
    hw = [(500, 375), (375, 500), (77, 50), (75, 50), (224, 224), (1, 9), (1000, 333)]
    plan, info = N.preprocess_plan(hw, 224)
    for (h, w), row in zip(hw, plan.tolist()):
        short, long = (w, h) if w <= h else (h, w)
        oh, ow = (int(224 * long / short), 224) if w <= h else (224, int(224 * long / short))
        assert row[1:5] == [h, w, oh, ow]
        assert row[5] == int(round((oh - 224) / 2.0)) and row[6] == int(round((ow - 224) / 2.0))
    assert info["max_h"] == 1000 and info["pixel_bytes"] == sum(3 * h * w for h, w in hw)
    assert plan[:, 0].tolist() == np.cumsum([0] + [3 * h * w for h, w in hw[:-1]]).tolist()
    assert info["ws_bytes"] > info["coef_bytes"] > 0 and info["coef_bytes"] % 16 == 0
    sq, _ = N.preprocess_plan(hw, 64, "squash", "bilinear")
    assert sq[:, 3:7].tolist() == [[64, 64, 0, 0]] * len(hw)
    with pytest.raises(ValueError):
        N.preprocess_plan([(0, 5)], 32)
    with pytest.raises(ValueError):
        N.preprocess_plan([(8, 8)], 32, "longest")


def test_device_preprocess_packs_ragged_batches_and_rejects_bad_input():
    from semanticlens_amd.foundation_models import DevicePreprocess

    pp = DevicePreprocess(size=32)
    a = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    b = np.full((4, 3), 9, np.uint8)  # grey image -> replicated channels
    buf, plan, info = pp.pack([a, torch.from_numpy(b)])
    assert info["pixel_bytes"] == a.size + 3 * b.size
    assert np.array_equal(buf.numpy()[: a.size], a.reshape(-1))
    assert np.all(buf.numpy()[a.size : a.size + 3 * b.size] == 9)
    assert plan[:, 1:3].tolist() == [[5, 7], [4, 3]]
    with pytest.raises(TypeError):
        pp.pack([a.astype(np.float32)])
    with pytest.raises(ValueError):
        pp.pack([np.zeros((4, 4, 4), np.uint8)])
    with pytest.raises(ValueError):
        DevicePreprocess(resize_mode="longest")
    if not torch.cuda.is_available():
        with pytest.raises(N.NativeLibraryError):  # no CPU fallback
            pp([a])


def test_visualize_components_renders_and_saves(mock_model, mock_dataset, tmp_path):
    """Host-side plotting (reference: activation_based.py:453-543): one tile per component from its top samples."""
    import matplotlib

    matplotlib.use("Agg")
    from semanticlens_amd.component_visualization.activation_based import _image_grid

    grid = _image_grid([torch.full((3, 4, 5), float(i)) for i in range(5)], per_row=3)
    assert grid.shape == (3, 2 * 6 + 2, 3 * 7 + 2)  # 2 rows x 3 columns of (4+2) x (5+2) cells + the closing frame
    assert grid[:, 2:6, 2:7].eq(0).all() and grid[:, 2:6, 9:14].eq(1).all() and grid[:, 8:12, 9:14].eq(4).all()
    assert grid[:, :2].eq(0).all() and grid[:, :, :2].eq(0).all()
    assert _image_grid([torch.ones(4, 5)], per_row=3).shape == (3, 8, 9)  # grey image -> 3 channels

    cv = ActivationComponentVisualizer(mock_model, mock_dataset, mock_dataset, ["0"], num_samples=4,
                                       aggregate_fn=agg.aggregate_conv_mean, cache_dir=tmp_path)
    am = cv.actmax_cache.cache["0"]
    am.n_latents, am.is_setup = 5, True
    am.sample_ids = torch.arange(20).reshape(5, 4) % len(mock_dataset)
    am.activations = torch.zeros(5, 4, dtype=torch.bfloat16)
    fig = cv.visualize_components(torch.tensor([0, 3, 4]), "0", n_samples=4, nrows=2, fname="demo")
    assert len(fig.axes) >= 3
    saved = list((cv.storage_dir / "plots").glob("demo_0_0-3-4.png"))
    assert len(saved) == 1 and saved[0].stat().st_size > 0
    with pytest.raises(ValueError):
        cv.visualize_components(torch.tensor([0]), "nope")


# ---- foundation_models.clip wrappers (reference tests/foundation_models/test_clip.py) with a stand-in open_clip -------
@pytest.fixture
def fake_open_clip(monkeypatch):
    """open_clip is third-party and not installed here: a stand-in module with the two entry points the wrappers use,
    recording how it was called.  The towers are tiny torch modules with CLIP's encode_* / context_length surface."""
    import types

    calls = []

    class Tiny(nn.Module):
        context_length = 7

        def __init__(self):
            super().__init__()
            self.img = nn.Linear(3 * 8 * 8, 12)
            self.txt = nn.Embedding(50, 12)

        def encode_image(self, x):
            return self.img(x.flatten(1))

        def encode_text(self, t):
            return self.txt(t).mean(1)

    def create_model_and_transforms(url, **kwargs):
        calls.append(("create", url, kwargs))
        to_tensor = lambda im: torch.from_numpy(np.asarray(im.resize((8, 8)), dtype=np.float32)).permute(2, 0, 1) / 255  # noqa: E731
        return Tiny(), None, to_tensor

    def get_tokenizer(url):
        calls.append(("tokenizer", url))
        return lambda txt, context_length: torch.stack(
            [torch.tensor(([ord(c) % 50 for c in s] + [0] * context_length)[:context_length]) for s in ([txt] if isinstance(txt, str) else txt)])

    mod = types.ModuleType("open_clip")
    mod.create_model_and_transforms, mod.get_tokenizer = create_model_and_transforms, get_tokenizer
    monkeypatch.setitem(sys.modules, "open_clip", mod)
    return calls


def test_open_clip_wrappers_shapes_and_plumbing(fake_open_clip):
    from PIL import Image

    from semanticlens_amd.foundation_models import ClipMobile, OpenClip, SigLipV2

    img, text = Image.new("RGB", (64, 64), color="red"), ["a red square", "a photo of a cat"]
    fm = OpenClip(url="ViT-B-32-quickgelu", device="cpu", load_weights=False)
    assert fake_open_clip[0] == ("create", "ViT-B-32-quickgelu", {"load_weights": False})
    assert "cpu" in str(fm.device) and "OpenClip(url='ViT-B-32-quickgelu'" in repr(fm)
    x = fm.preprocess(img)
    assert x.ndim == 4  # a single image gets a batch axis (clip.py:160-162)
    assert fm.preprocess([img, img]).shape[0] == 2
    fi, ft = fm.encode_image(x), fm.encode_text(fm.tokenize(text))
    assert fi.shape == (1, 12) and ft.shape == (2, 12) and not fi.requires_grad
    assert fm.tokenize(text).shape == (2, 7) and fm.tokenize(text, context_length=5).shape == (2, 5)
    for cls, url, extra in ((SigLipV2, "hf-hub:timm/ViT-B-16-SigLIP2", {}), (ClipMobile, "MobileCLIP-S1", {"pretrained": "datacompdr"})):
        del fake_open_clip[:]
        fm = cls(device="cpu", load_weights=False)
        assert fake_open_clip[0] == ("create", url, {**extra, "load_weights": False}) and fake_open_clip[1] == ("tokenizer", url)
        assert fm.encode_image(fm.preprocess(img)).shape[1] == fm.encode_text(fm.tokenize(text)).shape[1]
    assert ClipMobile(version="s2").url == "MobileCLIP-S2"


def test_open_clip_preprocess_keeps_a_batch_the_transform_returns(fake_open_clip):
    """clip.py:157-162: a non-list input goes through the preprocessor as is and only a 3-D result gains the batch axis; a
    transform that already returns (B, 3, S, S) must not be stacked into 5-D."""
    from PIL import Image

    from semanticlens_amd.foundation_models import OpenClip

    fm = OpenClip(url="x", device="cpu")
    img = Image.new("RGB", (16, 16))
    single = fm.preprocessor
    assert fm.preprocess(img).shape == (1, 3, 8, 8)
    fm.preprocessor = lambda im: torch.stack([single(im), single(im)])  # a batching transform
    assert fm.preprocess(img).shape == (2, 3, 8, 8)
    assert fm.preprocess([img, img, img]).shape == (3, 2, 3, 8, 8)  # the reference stacks whatever the transform gave, too


def test_open_clip_missing_raises_import_error():
    from semanticlens_amd.foundation_models import OpenClip

    if "open_clip" in sys.modules:
        pytest.skip("open_clip importable here")
    with pytest.raises(ImportError):
        OpenClip("ViT-B-32")


def test_device_preprocess_from_transform_reads_open_clip_pipeline():
    """from_transform introspects a torchvision-style Compose by class name (torchvision itself is not installed)."""
    from semanticlens_amd.foundation_models import DevicePreprocess

    def T(name, **attrs):
        return type(name, (), attrs)()

    class Interp:
        value = "bicubic"

    shortest = T("Compose", transforms=[T("Resize", size=224, interpolation=Interp()), T("CenterCrop", size=(224, 224)),
                                        T("_convert_to_rgb"), T("ToTensor"), T("Normalize", mean=(0.5, 0.4, 0.3), std=(0.2, 0.25, 0.3))])
    pp = DevicePreprocess.from_transform(shortest)
    assert (pp.size, pp.resize_mode, pp.interpolation, pp.mean, pp.std) == (224, "shortest", "bicubic", (0.5, 0.4, 0.3), (0.2, 0.25, 0.3))
    squash = T("Compose", transforms=[T("Resize", size=(256, 256), interpolation="bilinear"), T("ToTensor")])
    pp = DevicePreprocess.from_transform(squash)
    assert (pp.size, pp.resize_mode, pp.interpolation) == (256, "squash", "bilinear")
    with pytest.raises(ValueError):
        DevicePreprocess.from_transform(T("Compose", transforms=[T("ToTensor")]))
    with pytest.raises(ValueError):
        DevicePreprocess.from_transform(T("Compose", transforms=[T("Resize", size=224, interpolation=Interp()), T("CenterCrop", size=200)]))


def test_denormalization_transform_inverts_normalisation():
    from semanticlens_amd.utils import get_denormalization_transform

    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    img = torch.rand(3, 8, 9)
    norm = (img - mean[:, None, None]) / std[:, None, None]
    den = get_denormalization_transform()
    assert torch.allclose(den(norm), img, atol=1e-6)
    assert torch.allclose(den(norm[None].repeat(2, 1, 1, 1)), img[None].repeat(2, 1, 1, 1), atol=1e-6)
    assert torch.allclose(get_denormalization_transform([0.5] * 3, [0.25] * 3)(torch.zeros(3, 2, 2)), torch.full((3, 2, 2), 0.5))


# ---- utils.log_setup (reference tests/test_log_setup.py) ------------------------------------------------------------
def test_setup_colored_logging_levels_env_and_file(monkeypatch, tmp_path, caplog):
    import logging

    from semanticlens_amd.utils import setup_colored_logging
    from semanticlens_amd.utils.log_setup import PACKAGE, ColorFormatter

    logger = logging.getLogger(PACKAGE)
    monkeypatch.delenv("SEMANTICLENS_LOG_LEVEL", raising=False)
    setup_colored_logging()
    assert logger.level == logging.INFO
    with caplog.at_level(logging.INFO, logger=PACKAGE):
        logger.debug("hidden")
        logger.info("shown")
    assert "shown" in caplog.text and "hidden" not in caplog.text
    setup_colored_logging(log_level="DEBUG")
    assert logger.level == logging.DEBUG and len(logger.handlers) == 1  # handlers are replaced, not stacked
    monkeypatch.setenv("SEMANTICLENS_LOG_LEVEL", "WARNING")
    setup_colored_logging()  # the environment variable wins over the argument
    assert logger.level == logging.WARNING
    monkeypatch.delenv("SEMANTICLENS_LOG_LEVEL")
    log_file = tmp_path / "test.log"
    setup_colored_logging(file_path=str(log_file))
    logger.warning("to the file")
    for h in logger.handlers:
        h.flush()
    assert "to the file" in log_file.read_text() and "\\033[" not in log_file.read_text()
    rec = logging.LogRecord("t", logging.WARNING, "/fake/path.py", 10, "A test warning", (), None)
    out = ColorFormatter("[%(levelname)s]: %(message)s", use_color=True).format(rec)
    assert out.startswith(ColorFormatter.COLOR_MAP["WARNING"]) and out.endswith(ColorFormatter.RESET_SEQ) and rec.short_filename == "path.py"
    assert ColorFormatter("[%(levelname)s]: %(message)s", use_color=False).format(rec) == "[WARNING]: A test warning"
    setup_colored_logging("nonsense")
    assert logger.level == logging.INFO
    for h in list(logger.handlers):
        logger.removeHandler(h)
    logger.addHandler(logging.NullHandler())


def test_flatten_spatial_never_uses_the_stride_of_a_size_one_dimension():
    """Host half of K1: (B,C,H,W) -> (sb, sc, ss) of the (B,C,H*W) view.  torch keeps an arbitrary stride for a size-1
    dimension (a transposed (B,C,1,W) is "contiguous" with stride 8 on its last axis); using it gave wrong maxima."""
    from semanticlens_amd._native import _flatten_spatial

    def walk(x):
        v, sb, sc, ss = _flatten_spatial(x)
        B, C, H, W = x.shape
        flat = v.as_strided((B, C, H * W), (sb, sc, ss), v.storage_offset())
        return flat

    base = torch.arange(4 * 3 * 8, dtype=torch.float32).reshape(4, 3, 1, 8)
    for x in (base, base.transpose(2, 3), base.transpose(2, 3)[:, :, ::2], base[:, :, :, ::3],
              torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5).transpose(2, 3),
              torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5)[:, :, 1:3, 1:4],
              torch.arange(2 * 3 * 4 * 5, dtype=torch.float32).reshape(2, 3, 4, 5).contiguous(memory_format=torch.channels_last)):
        assert torch.equal(walk(x).amax(-1), x.amax((2, 3))), tuple(x.shape)


def test_pinned_stack_falls_back_to_default_collate_without_a_device():
    from semanticlens_amd.component_visualization._prefetch import PinnedStack

    stack = PinnedStack()
    batch = [(torch.full((2, 3), float(i)), i) for i in range(5)]
    out = stack(batch)
    if not torch.cuda.is_available():
        assert stack.last_slot == -1
    assert out.shape == (5, 2, 3) and torch.equal(out[:, 0, 0], torch.arange(5.0))
    ragged = [(torch.zeros(2), 0), (torch.zeros(3), 1)]
    with pytest.raises(RuntimeError):
        stack(ragged)  # default_collate's own error, as in the reference


def test_native_block_refuses_head_dims_the_attention_kernels_are_not_built_for():
    """`_HEAD_DIMS` listed every multiple of 8 while encoder.hip instantiates eight of them; a head_dim-48 model
    constructed, split all its weights and died on the first encode.  Now refused at construction."""
    import re

    from semanticlens_amd.foundation_models import native_clip as nc

    src = (ROOT / "semanticlens_amd" / "csrc" / "encoder.hip").read_text()
    for fn in ("launch_attention_mfma", "launch_attention_bf16x3"):
        built = sorted({int(m) for m in re.findall(r"case (\d+): return %s<\1>" % fn, src)})
        assert tuple(built) == nc._HEAD_DIMS, (fn, built)
    for width, heads in ((192, 4), (320, 8), (448, 8), (896, 8), (960, 8)):  # head_dim 48, 40, 56, 112, 120
        with pytest.raises(ValueError, match="head_dim"):
            nc._BlockWeights(width, heads, 0)
    assert nc._BlockWeights(1152, 16, 0).head_dim == 72


def test_embed_stage_writes_rows_in_dataset_order_when_preprocess_output_types_mix(monkeypatch):
    """a non-tensor `preprocess` result was encoded at once while earlier tensor batches were still held back,
    so its rows landed in front of theirs."""
    from semanticlens_amd.component_visualization import activation_based as ab

    monkeypatch.setattr(ab.N, "to_device", lambda t, device=None: t)

    class FM:
        embed_accumulate = 8

        def preprocess(self, items):
            return items

        def encode_image(self, pre):
            x = pre if torch.is_tensor(pre) else torch.stack(list(pre))
            return x.reshape(x.shape[0], -1)[:, :1].float() * torch.ones(1, 3)

    stage = ab._EmbedStage(FM(), 7, batch_hint=2)
    stage.add(None, torch.tensor([[0.0], [1.0]]))
    stage.add(None, torch.tensor([[2.0], [3.0]]))
    stage.add(None, [torch.tensor([4.0]), torch.tensor([5.0])])  # a list: cannot be held back
    stage.add(None, torch.tensor([[6.0]]))
    out = stage.finish()
    assert out[:, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0]


def test_friendly_batch_fills_whole_tile_rounds():
    """`NativeSigLip.embed_accumulate`: the image count whose token rows make every GEMM of a block a whole number of 256-tile rounds."""
    from semanticlens_amd.foundation_models.native_clip import friendly_batch

    def rounds_per_image(b, tokens, width, mlp, cus=256):
        tm = -(-b * tokens // 256)
        return sum(-(-tm * tn // cus) for tn in (-(-3 * width // 256), -(-width // 256), -(-mlp // 256), -(-width // 256))) / b

    b = friendly_batch(256, 1152, 4304)  # SigLIP-so400m at 224 px
    assert 64 <= b <= 256 and b % 8 == 0
    assert rounds_per_image(b, 256, 1152, 4304) <= min(rounds_per_image(c, 256, 1152, 4304) for c in (64, 128, 192, 256)) + 1e-12
    assert rounds_per_image(b, 256, 1152, 4304) < 0.85 * rounds_per_image(64, 256, 1152, 4304)  # what 64 images wasted
    assert 64 <= friendly_batch(729, 1152, 4304) <= 256 and 64 <= friendly_batch(196, 768, 3072) <= 256


def test_text_chunks_are_tokenised_one_ahead_in_the_same_order(monkeypatch):
    """`_encode_texts`: the tokenizer runs one chunk ahead on a helper thread; calls, order and results equal the serial loop's."""
    from semanticlens_amd.lens import _encode_texts

    class FM:
        device = torch.device("cpu")

        def __init__(self):
            self.log = []

        def tokenize(self, chunk):
            self.log.append(("tok", tuple(chunk)))
            return torch.tensor([[len(t), sum(map(ord, t)) % 97] for t in chunk], dtype=torch.int64)

        def encode_text(self, tokens):
            self.log.append(("enc", tokens.shape[0]))
            return tokens.float() * 0.5

    texts = [f"prompt {i}" * (1 + i % 3) for i in range(23)]
    a, b = FM(), FM()
    monkeypatch.setenv("SL_TEXT_PREFETCH", "1")
    got = _encode_texts(a, texts, batch_size=5)
    monkeypatch.setenv("SL_TEXT_PREFETCH", "0")
    want = _encode_texts(b, texts, batch_size=5)
    assert torch.equal(got, want) and got.shape == (23, 2)
    assert [e for e in a.log if e[0] == "tok"] == [e for e in b.log if e[0] == "tok"]  # same chunks, same order
    assert [e for e in a.log if e[0] == "enc"] == [("enc", 5)] * 4 + [("enc", 3)]
    assert len(_encode_texts(FM(), texts, batch_size=None)) == 23  # one chunk: no thread


def test_reduce_policy_tuner_measures_every_candidate_and_keeps_the_default_unless_clearly_slower(monkeypatch):
    """`N.ReducePolicyTuner`: one untimed + three timed launches per candidate, decision from the medians, the library default unless the
    other policy is > 2 % faster; off for small inputs and once the caller chose a policy."""
    applied = []
    monkeypatch.delenv("SL_NT_MIN_BYTES", raising=False), monkeypatch.delenv("SL_REDUCE_TAIL_MB", raising=False)
    monkeypatch.setattr(N, "_policy_explicit", False)
    clock = {"now": 0.0, "cost": {}}

    class Ev:
        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = clock["now"]

        def query(self):
            return True

        def elapsed_time(self, other):
            return other.t - self.t

    monkeypatch.setattr(torch.cuda, "Event", Ev)

    # a launch costs what the policy in force says
    current = [(None, None)]
    monkeypatch.setattr(N, "_set_reduce_policy_raw", lambda a, b: (applied.append((a, b)), current.__setitem__(0, (a, b))))
    cands = N.ReducePolicyTuner.CANDIDATES
    assert cands[0] == (None, None) and len(cands) == 3
    n_trials = 4 * len(cands)  # one untimed + three timed launches per candidate

    def go(tuner, costs, n=n_trials + 4, nbytes=200 << 20):
        clock["cost"] = dict(zip(cands, costs))
        for _ in range(n):
            tuner.run(lambda: clock.__setitem__("now", clock["now"] + clock["cost"][current[0]]), nbytes, 256)

    t = N.ReducePolicyTuner()
    go(t, (1.0, 0.9, 0.95))
    assert t.choice == 1 and applied.count(cands[1]) >= 4 + 4  # trials, then every launch under the chosen policy
    assert applied[-1] == (None, None)  # always restored
    t = N.ReducePolicyTuner()
    go(t, (1.0, 0.95, 0.9))
    assert t.choice == 2  # the 128 MiB tail (round 5: outputs of three-stream residual adds)
    t = N.ReducePolicyTuner()
    go(t, (1.0, 0.99, 0.985))
    assert t.choice == 0  # within 2 %: the default stays
    t = N.ReducePolicyTuner()
    before = len(applied)
    go(t, (1.0, 0.5, 0.5), nbytes=50 << 20)
    assert t.choice is None and len(applied) == before  # below 96 MiB: never touched
    monkeypatch.setattr(N, "_policy_explicit", True)
    t = N.ReducePolicyTuner()
    go(t, (1.0, 0.5, 0.5))
    assert t.choice is None and len(applied) == before  # the caller's explicit policy wins
    assert N.ReducePolicyTuner.for_site(("m", "layer")) is N.ReducePolicyTuner.for_site(("m", "layer"))


# ---- the aten tie order's host self-test: restatement vs the installed torch.topk, no device -----------------------
def test_aten_order_restatement_equals_the_installed_torch_topk():
    """`sl_aten_topk_order_host` is host code: the positions the K3 kernels' restatement selects against `torch.topk` on the CPU of
    THIS container (torch 2.10 / libstdc++ 11), both ATen branches, NaN, the -0.0 sentinel row."""
    res = N.aten_order_selftest(rows=400, force=True)
    assert res["rows"] == 400 and res["mismatches"] == 0, res
    g = torch.Generator().manual_seed(77)
    for k, n in ((20, 276), (100, 164), (20, 1300), (1, 64), (7, 7), (100, 6600)):  # k * 64 <= n: partial_sort — 1300, 64, 6600
        for _ in range(25):
            row = (torch.randint(-3, 9, (n,), generator=g).float() / 2).to(torch.bfloat16)
            assert torch.equal(N.aten_topk_order_host(row, k), torch.topk(row, k).indices.to(torch.int32)), (k, n)
    with pytest.raises(ValueError, match="sl_aten_topk_order_host"):
        N.aten_topk_order_host(torch.zeros(4, dtype=torch.bfloat16), 5)


def test_aten_order_selftest_warns_on_a_host_with_another_tie_order(monkeypatch):
    """A torch whose topk breaks ties differently (here: a stable sort, lowest position first) is reported once, with versions."""
    real = torch.topk

    def stable_topk(x, k, *a, **kw):
        order = torch.sort(x.float().nan_to_num(nan=float("inf")), descending=True, stable=True).indices[:k]
        return type("R", (), {"indices": order, "values": x[order]})()

    monkeypatch.setattr(torch, "topk", stable_topk)
    with pytest.warns(RuntimeWarning, match="selects different positions on"):
        res = N.aten_order_selftest(force=True)
    assert res["mismatches"] > 0
    monkeypatch.setattr(torch, "topk", real)
    assert N.aten_order_selftest(force=True)["mismatches"] == 0
