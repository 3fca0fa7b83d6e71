"""Sharded concept-DB build on the device: 2 and 3 ranks SHARING one GPU (gloo, collectives staged through the
host) must reproduce the single-process total-order result bit for bit — real K1/K3 per shard, real K4 merge,
real sharded K5 gather; only the transport differs from production (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(tie_mode="total"):
    from helpers import FakeVLM, TensorPairDataset, make_int_conv_model, make_int_images
    from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators

    model = make_int_conv_model().to("cuda:0")
    ds = TensorPairDataset(make_int_images(47))  # 47: shards of unequal size
    cv = ActivationComponentVisualizer(model, ds, ds, ["0", "2"], num_samples=6,
                                       aggregate_fn=aggregators.aggregate_conv_max, tie_mode=tie_mode)
    return cv, FakeVLM().to("cuda:0")


def _worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from semanticlens_amd import distributed as sld

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cv, fm = _build()
        db = sld.compute_concept_db_sharded(cv, fm, batch_size=8)
        cv2, fm2 = _build()
        db_ref = sld.compute_concept_db_sharded(cv2, fm2, batch_size=8, referenced_only=True)
        assert fm2.calls["encode_image"] <= fm.calls["encode_image"]
        np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"),
                 **{f"db_{k}": v.cpu().numpy() for k, v in db.items()},
                 **{f"dbref_{k}": v.cpu().numpy() for k, v in db_ref.items()},
                 **{f"ids_{k}": cv.get_max_reference(k).numpy() for k in db},
                 **{f"vals_{k}": cv.actmax_cache.cache[k].activations.view(torch.int16).numpy() for k in db})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_build_equals_single_process(world, tmp_path):
    cv, fm = _build()
    want = cv._compute_concept_db(fm, batch_size=8)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"w{world}_r{r}.npz")
        for k in ("0", "2"):
            assert np.array_equal(got[f"ids_{k}"], cv.get_max_reference(k).numpy()), (world, r, k)
            assert np.array_equal(got[f"vals_{k}"], cv.actmax_cache.cache[k].activations.view(torch.int16).numpy())
            assert np.array_equal(got[f"db_{k}"], want[k].numpy()), (world, r, k)
            assert np.array_equal(got[f"dbref_{k}"], want[k].numpy()), (world, r, k, "referenced_only")


def test_sharded_requires_total_order():
    from semanticlens_amd import distributed as sld

    cv, fm = _build("aten")
    with pytest.raises(ValueError, match="tie_mode='total'"):
        sld.run_sharded(cv)


# ---- analysis stage sharded: native text tower + cosine GEMM + scores per rank, one all-gather of result rows ---------
def _analysis_inputs():
    import synth
    from semanticlens_amd.foundation_models.native_clip import NativeClip

    base = synth.SyntheticClip(device="cuda:0", seed=2, embed_dim=64, image_size=64, patch=16, v_width=128, v_layers=1,
                               v_heads=2, t_width=128, t_layers=2, t_heads=2, vocab=49408)
    fm = NativeClip(base)
    g = torch.Generator().manual_seed(9)
    db = {"a": torch.randn(96, 64, generator=g).to("cuda:0"), "q": torch.randn(7, 64, generator=g).to("cuda:0")}  # "q": C == Q quirk
    V = torch.randn(11, 6, 64, generator=g).to("cuda:0")
    queries = ["red car", "a dog", "striped zebra in grass", "sky", "wheel", "two cats", "x"]
    templates = ["a photo of a {}", "an image of {}"]
    return fm, db, V, queries, templates


def _analysis_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    from semanticlens_amd import distributed as sld
    from semanticlens_amd import scores

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fm, db, V, queries, templates = _analysis_inputs()
        out = {}
        for tag, tpl in (("plain", None), ("tpl", templates)):
            for k, v in sld.text_probing_sharded(fm, queries, db, templates=tpl, batch_size=3).items():
                out[f"{tag}_{k}"] = v.cpu().numpy()
        out["clarity"] = sld.eval_sharded(scores.clarity_score, V).cpu().numpy()
        out["poly"] = sld.eval_sharded(scores.polysemanticity_score, {"x": V})["x"].cpu().numpy()
        np.savez(os.path.join(out_dir, f"an_w{world}_r{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_probing_and_scores_equal_single_process(world, tmp_path):
    from semanticlens_amd import Lens, scores

    fm, db, V, queries, templates = _analysis_inputs()
    lens = Lens(fm, device="cuda:0")
    want = {}
    for tag, tpl in (("plain", None), ("tpl", templates)):
        for k, v in lens.text_probing(queries, db, templates=tpl, batch_size=3).items():
            want[f"{tag}_{k}"] = v.cpu().numpy()
    want["clarity"] = scores.clarity_score(V).cpu().numpy()
    want["poly"] = scores.polysemanticity_score(V).cpu().numpy()
    mp.spawn(_analysis_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"an_w{world}_r{r}.npz")
        for k, v in want.items():
            assert got[k].shape == v.shape, (world, r, k)
            assert np.array_equal(got[k], v), (world, r, k, np.abs(got[k] - v).max())
