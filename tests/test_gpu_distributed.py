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


def _build(tie_mode="total", cache_dir=None, device="cuda:0"):
    from helpers import FakeVLM, TensorPairDataset, make_int_conv_model, make_int_images
    from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators

    model = make_int_conv_model().to(device)
    ds = TensorPairDataset(make_int_images(47))  # 47: shards of unequal size
    cv = ActivationComponentVisualizer(model, ds, ds, ["0", "2"], num_samples=6, cache_dir=cache_dir,
                                       aggregate_fn=aggregators.aggregate_conv_max, tie_mode=tie_mode)
    return cv, FakeVLM().to(device)


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


@pytest.mark.parametrize("world", [2, 3, 8])  # 8 = BASELINE configs[2]'s rank count (47 samples: shards of 6, the last one 5)
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


def _cache_worker(rank, world, port, out_dir, cache_dir):
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from semanticlens_amd import distributed as sld

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cv, fm = _build(cache_dir=cache_dir)  # the constructor loads an existing top-k cache into actmax_cache
        sld.run_sharded(cv, batch_size=8)
        np.savez(os.path.join(out_dir, f"cache_r{rank}.npz"),
                 **{f"ids_{k}": cv.get_max_reference(k).numpy() for k in ("0", "2")},
                 **{f"vals_{k}": cv.actmax_cache.cache[k].activations.view(torch.int16).numpy() for k in ("0", "2")})
    finally:
        dist.destroy_process_group()


def test_sharded_run_with_cache_dir(tmp_path):
    """with `cache_dir` set the constructor may already have loaded a top-k cache; the sharded run must
    neither collect on top of it (duplicate ids) nor skip storing its merged result."""
    cv, _ = _build()
    cv.run(batch_size=8)
    want = {k: (cv.get_max_reference(k).numpy(), cv.actmax_cache.cache[k].activations.view(torch.int16).numpy()) for k in ("0", "2")}
    cache = tmp_path / "cache"
    out = tmp_path / "out"
    out.mkdir()

    def check():
        for r in range(2):
            got = np.load(out / f"cache_r{r}.npz")
            for k in ("0", "2"):
                assert np.array_equal(got[f"ids_{k}"], want[k][0]), (r, k)
                assert np.array_equal(got[f"vals_{k}"], want[k][1]), (r, k)
                ids = got[f"ids_{k}"]
                assert all(len(set(row[row >= 0])) == (row >= 0).sum() for row in ids), "a sample id is listed twice"

    mp.spawn(_cache_worker, args=(2, _free_port(), str(out), str(cache)), nprocs=2, join=True)  # miss: collect + store
    check()
    stored = sorted(p.name for p in cache.rglob("*.safetensors"))
    assert len(stored) == 2, stored  # rank 0 stored the merged states, one file per layer
    cv1, _ = _build(cache_dir=str(cache))  # a single-process visualizer reads what the sharded run wrote
    for k in ("0", "2"):
        assert np.array_equal(cv1.get_max_reference(k).numpy(), want[k][0])
    mp.spawn(_cache_worker, args=(2, _free_port(), str(out), str(cache)), nprocs=2, join=True)  # hit on every rank
    check()
    # a partial cache (one layer's file missing) is a miss: states loaded by the constructor must not be collected over
    next(cache.rglob("*-2.safetensors")).unlink()
    mp.spawn(_cache_worker, args=(2, _free_port(), str(out), str(cache)), nprocs=2, join=True)
    check()


def _bench_line(extra_env, args, nproc):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(nproc)] + args
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


BUILD_ONLY = ["--no-tokens-leg", "--no-config4-leg", "--no-probing", "--no-cpu-baseline"]  # the headline job (+ strong scaling) alone


def test_bench_two_ranks_sharing_one_gpu_over_gloo():
    """BASELINE configs[2] code path (`bench.py --gpus N`): per-rank collect, packed all-gather, K4 merge, sharded K5
    gather + all-reduce, max-over-ranks timing — with two ranks on ONE GPU and gloo as the transport.  The N > 1 line is COMPLETE:
    configs[3] / configs[4] collect legs with the cross-rank merge, text_probing with the query rows sharded, the scores with the
    component axis sharded, the CPU baseline (rank 0) and the oracle / single-process parity of each."""
    line = _bench_line({"SL_BENCH_BACKEND": "gloo", "SL_BENCH_SHARE_GPU": "1"}, ["--steps", "3", "--batches-per-step", "2", "--warmup", "1", "--batch", "64", "--min-warmup-seconds", "0.3",
                                                                                 "--strong-images", "1500", "--strong-pool-batches", "4", "--cpu-images", "64", "--leg-steps", "2"], 2)
    assert line["strong_scaling"]["images"] == 1500 and line["strong_scaling"]["tie_mode"] == "total"  # (the default is the 1.28 M-image job: 200 s here)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["images_total"] == 2 * 3 * 2 * 64 and line["value"] > 0
    chk = line["sharded_check"]
    assert chk["topk_values_equal"] and chk["topk_ids_equal"] and chk["concept_db_equal"] and chk["probe_equal"] and chk["scores_equal"], chk
    tp = line["text_probing"]
    assert tp["n_gpus"] == 2 and tp["value"] > 0 and tp["roofline"]["frac"] > 0 and tp["compute_only"]["value"] >= tp["value"] * 0.5
    assert tp["sharded_check"]["equals_single_process_bitwise"] and tp["sharded_check"]["max_abs_diff_vs_oracle_64_queries"] < 1e-4
    assert tp["from_prompts"]["n_gpus"] == 2 and tp["from_prompts"]["queries_per_s"] > 0
    c3, c4 = line["config3_full"], line["config4_full"]
    assert c3["n_gpus"] == 2 and c3["images"] == 2 * 2 * 64 and c3["self_check"] == "ok" and c3["roofline"]["frac"] > 0
    assert c3["text_probing_from_prompts"]["n_gpus"] == 2 and c3["text_probing_from_prompts"]["max_abs_diff_vs_oracle_64_queries"] < 1e-4
    assert c4["n_gpus"] == 2 and c4["images"] == 2 * 2 * 64 and c4["self_check"] == "ok"
    assert c4["relevance_visualizer"]["n_gpus"] == 2 and c4["relevance_visualizer"]["images"] == 256
    sc = c4["scores_full_db"]
    assert sc["n_gpus"] == 2 and sc["sharded_equals_single_process"] is True and all(v < 1e-4 for v in sc["max_abs_diff_vs_oracle"].values())
    cpu = line["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["all_cores"]["value"] > 0 and cpu["kind"] == "port"
    line = _bench_line({"SL_BENCH_BACKEND": "gloo", "SL_BENCH_SHARE_GPU": "1"},
                       ["--scaling", "strong", "--images", "500", "--batches-per-step", "1", "--warmup", "1", "--batch", "64", "--min-warmup-seconds", "0"] + BUILD_ONLY, 2)
    assert line["scaling"] == "strong" and line["config"]["images_total"] == 500 and line["steps"] == 4


def _nccl_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from semanticlens_amd import distributed as sld

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        cv, fm = _build(device=dev)
        db = sld.compute_concept_db_sharded(cv, fm, batch_size=8)
        np.savez(os.path.join(out_dir, f"nccl_r{rank}.npz"), **{f"db_{k}": v.cpu().numpy() for k, v in db.items()},
                 **{f"ids_{k}": cv.get_max_reference(k).numpy() for k in db})
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the RCCL (nccl backend) branch of distributed.py")
def test_sharded_build_over_rccl(tmp_path):
    """One process per GPU over RCCL — runs on the first multi-GPU box that executes the suite."""
    world = min(torch.cuda.device_count(), 8)
    cv, fm = _build()
    want = cv._compute_concept_db(fm, batch_size=8)
    mp.spawn(_nccl_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"nccl_r{r}.npz")
        for k in ("0", "2"):
            assert np.array_equal(got[f"ids_{k}"], cv.get_max_reference(k).numpy()), (r, k)
            assert np.array_equal(got[f"db_{k}"], want[k].numpy()), (r, k)


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


# ---- the RCCL transport on ONE GPU: a world_size=1 "nccl" process group ------------------------------------------------
def _nccl_single_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    from semanticlens_amd import distributed as sld
    from semanticlens_amd import scores

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        assert dist.get_backend() == "nccl" and not sld._host_staged(None)
        cv, fm = _build(device=dev)
        db = sld.compute_concept_db_sharded(cv, fm, batch_size=8)
        cv2, fm2 = _build(device=dev)
        db_ref = sld.compute_concept_db_sharded(cv2, fm2, batch_size=8, referenced_only=True)
        # the collectives themselves, on device buffers: packed-state all-gather, sharded gather + all-reduce, row all-gather
        states = [cv.actmax_cache.cache[k].device_state(dev) for k in ("0", "2")]
        gathered = sld.all_gather_states(states)
        assert all(g[0].is_cuda and g[0].shape[0] == 1 for g in gathered)
        for (v, i), (gv, gi) in zip(states, gathered):
            assert torch.equal(gv[0].view(torch.int16), v.view(torch.int16)) and torch.equal(gi[0], i)
        rows = torch.arange(35, dtype=torch.float32, device=dev).reshape(7, 5)
        assert torch.equal(sld.all_gather_rows(rows, 7), rows)
        fm_a, db_a, V, queries, templates = _analysis_inputs()
        out = {f"tpl_{k}": v.cpu().numpy() for k, v in sld.text_probing_sharded(fm_a, queries, db_a, templates=templates, batch_size=3).items()}
        out["clarity"] = sld.eval_sharded(scores.clarity_score, V).cpu().numpy()
        np.savez(os.path.join(out_dir, "nccl1.npz"), **{f"db_{k}": v.cpu().numpy() for k, v in db.items()},
                 **{f"dbref_{k}": v.cpu().numpy() for k, v in db_ref.items()},
                 **{f"ids_{k}": cv.get_max_reference(k).numpy() for k in db}, **out)
        if sld.COLLECTIVES == "native":
            # an index-less device ("cuda" after torch.cuda.set_device(rank), the usual pattern) names the current device: the
            # library's communicator must accept tensors living on cuda:0 (round-4 advisor finding)
            sld.destroy_native_comms()
            comm = sld.native_comm(None, "cuda")
            assert comm is not None and comm.device == dev and sld.native_comm(None, dev) is comm
            assert sld._all_reduce_host_ints([3, -5], dist.ReduceOp.MAX, None, "cuda") == [3, -5]
            assert torch.equal(sld.all_gather_rows(rows, 7), rows)
            from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators
            from helpers import TensorPairDataset, make_int_conv_model, make_int_images

            ds = TensorPairDataset(make_int_images(47))
            cv3 = ActivationComponentVisualizer(make_int_conv_model().to(dev), ds, ds, ["0", "2"], num_samples=6, device="cuda",
                                                aggregate_fn=aggregators.aggregate_conv_max, tie_mode="total")
            sld.run_sharded(cv3, batch_size=8)
            for k in ("0", "2"):
                assert torch.equal(cv3.get_max_reference(k), cv.get_max_reference(k))
    finally:
        dist.destroy_process_group()
    # a re-initialised process group must not be handed the communicator of the destroyed one
    if sld.COLLECTIVES != "native":
        return
    stale = comm
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, init_method=f"tcp://127.0.0.1:{_free_port()}")  # one rank: no port to agree on
    try:
        fresh = sld.native_comm(None, dev)
        assert fresh is not None and fresh is not stale and stale._h is None, "stale communicator survived destroy_process_group"
        assert sld._all_reduce_host_ints([7], dist.ReduceOp.SUM, None, dev) == [7]
    finally:
        sld.destroy_native_comms()
        dist.destroy_process_group()


def test_sharded_build_over_rccl_world_size_one(tmp_path):
    """The device-buffer branch of `distributed.py` (`all_gather_states`, `gather_concept_db_sharded`, `all_gather_rows`:
    collectives issued on HBM tensors, no host staging) under the production backend "nccl" (= RCCL) with a one-rank
    group — what a 1-GPU box can execute of configs[2]'s transport.  Results equal the plain single-process build."""
    from semanticlens_amd import Lens, scores

    cv, fm = _build()
    want = cv._compute_concept_db(fm, batch_size=8)
    mp.spawn(_nccl_single_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "nccl1.npz")
    for k in ("0", "2"):
        assert np.array_equal(got[f"ids_{k}"], cv.get_max_reference(k).numpy()), k
        assert np.array_equal(got[f"db_{k}"], want[k].numpy()), k
        assert np.array_equal(got[f"dbref_{k}"], want[k].numpy()), (k, "referenced_only")
    fm_a, db_a, V, queries, templates = _analysis_inputs()
    probe = Lens(fm_a, device="cuda:0").text_probing(queries, db_a, templates=templates, batch_size=3)
    for k, v in probe.items():
        assert np.array_equal(got[f"tpl_{k}"], v.cpu().numpy()), k
    assert np.array_equal(got["clarity"], scores.clarity_score(V).cpu().numpy())


def test_bench_distributed_path_over_rccl_on_one_gpu():
    """`bench.py` with SL_BENCH_FORCE_DIST=1: a one-rank nccl group, so the N > 1 code of the bench (barriers, packed all-gather,
    K4 merge, sharded K5 + all-reduce, max-over-ranks timing) runs over RCCL on the 1-GPU box."""
    line = _bench_line({"SL_BENCH_FORCE_DIST": "1"}, ["--steps", "2", "--batches-per-step", "2", "--warmup", "1", "--batch", "64",
                                                        "--min-warmup-seconds", "0.3", "--quick"], 1)
    assert line["n_gpus"] == 1 and line["config"]["collectives"].startswith("libsemanticlens_hip.so (RCCL behind the C ABI)")
    assert line["config"]["rccl_world_size"] == 1 and line["config"]["tie_mode"] == "total"
    assert all(line["sharded_check"][k_] for k_ in ("topk_values_equal", "topk_ids_equal", "concept_db_equal"))
    assert line["config"]["images_total"] == 2 * 2 * 64 and line["value"] > 0 and line["self_check"] == "ok"
    line = _bench_line({"SL_BENCH_FORCE_DIST": "1", "SL_COLLECTIVES": "torch"}, ["--steps", "1", "--batches-per-step", "2", "--warmup", "1",
                                                                                 "--batch", "64", "--min-warmup-seconds", "0", "--quick"], 1)
    assert line["config"]["collectives"].startswith("torch.distributed (nccl)") and line["config"]["rccl_world_size"] == 1


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts N ranks itself (torch.distributed.run
    on a free local port) and prints ONE line — here two ranks on the one GPU of the box (SL_BENCH_SHARE_GPU=1 -> gloo), with the
    strong-scaling job behind the weak-scaling one."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SL_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--batches-per-step", "2", "--warmup", "1",
           "--batch", "64", "--min-warmup-seconds", "0.2", "--strong-images", "333", "--strong-pool-batches", "2"] + BUILD_ONLY
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["images_total"] == 2 * 2 * 2 * 64
    assert line["config"]["tie_mode"] == "total" and line["config"]["process_group"] == {"backend": "gloo", "world_size": 2}
    chk = line["sharded_check"]  # two ranks' batches merged across the (gloo) wire == the oracle's replay on rank 0
    assert chk["ranks"] == 2 and chk["images"] == 128 and chk["topk_values_equal"] and chk["topk_ids_equal"] and chk["concept_db_equal"], chk
    st = line["strong_scaling"]
    assert st["images"] == 333 and st["n_gpus"] == 2 and st["images_per_gpu"] == 167 and st["seconds"] > 0
    # a speed-up is only ever quoted against an N = 1 record of the SAME job (image count, tie order): none exists for 333 images
    assert st["tie_mode"] == "total" and st["speedup_vs_n1"] is None and st["n1_reference"] is None
    assert line["k3"]["tie_mode"] == "total" and line["k3"]["launches"] > 0 and line["k3"]["avg_launch_us"] > 0
    # without the test switch a box with fewer GPUs than ranks is refused, loudly
    env.pop("SL_BENCH_SHARE_GPU")
    if torch.cuda.device_count() < 2:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
        assert res.returncode != 0 and "one rank per GPU" in (res.stderr + res.stdout)


# ---- RCCL behind the C ABI (csrc/comm.hip): sl_comm_*, sl_actmax_pack / _merge_packed / _allgather_merge -----------------
def _random_states(R, shapes, k, seed):
    """Per-rank total-order states of `shapes` layers over disjoint id ranges (oracle-built), tie-heavy values."""
    import oracle

    rng = np.random.RandomState(seed)
    per_rank, n = [], 70
    acts = [(rng.randint(-4, 40, size=(R * n, C)) / 8.0).astype(np.float32) for C in shapes]
    for r in range(R):
        states = []
        for a in acts:
            o = oracle.ActMaxOracle(k, a.shape[1], oracle.MODE_TOTAL)
            if r != 1:  # rank 1 of every test saw nothing: it contributes the initial state
                o.update(a[r * n:(r + 1) * n], np.arange(r * n, (r + 1) * n))
            states.append((o.vals.copy(), o.ids.copy()))
        per_rank.append(states)
    want = []
    for a in acts:
        o = oracle.ActMaxOracle(k, a.shape[1], oracle.MODE_TOTAL)
        keep = np.concatenate([np.arange(r * n, (r + 1) * n) for r in range(R) if r != 1])
        o.update(a[keep], keep)
        want.append((o.vals, o.ids))
    return per_rank, want


def _to_dev(states):
    return [(torch.from_numpy(v.view(np.int16)).view(torch.bfloat16).cuda(), torch.from_numpy(i).cuda()) for v, i in states]


@pytest.mark.parametrize("R,k", [(2, 6), (4, 20), (8, 100), (3, 1)])
def test_packed_merge_of_gathered_states_equals_the_oracle(R, k):
    """`sl_actmax_pack` + `sl_actmax_merge_packed` — the two local halves of `sl_actmax_allgather_merge` — on a gathered buffer
    assembled by hand: every rank ends with the oracle's global top-k, whichever block it leaves out as its own."""
    from semanticlens_amd import _native as N
    from semanticlens_amd import distributed as sld

    shapes = (37, 5, 130)  # odd sizes: the blocks' value sections are only 2-byte aligned
    per_rank, want = _random_states(R, shapes, k, seed=R * 100 + k)
    blocks = [N.actmax_pack(_to_dev(s)) for s in per_rank]
    for r in range(R):  # the packed layout is distributed.pack_states' (what the torch transport has always sent)
        ref = sld.pack_states([(torch.from_numpy(v.view(np.int16)).view(torch.bfloat16), torch.from_numpy(i)) for v, i in per_rank[r]])
        assert torch.equal(blocks[r].cpu(), ref), r
    gathered = torch.stack(blocks)
    for r in range(R):
        mine = _to_dev(per_rank[r])
        N.actmax_merge_packed(mine, gathered, skip_rank=r)
        for (v, i), (wv, wi) in zip(mine, want):
            assert np.array_equal(v.view(torch.int16).cpu().numpy().view(np.uint16), wv), (R, k, r)
            assert np.array_equal(i.cpu().numpy(), wi), (R, k, r)
    # skip_rank = -1: every block is foreign (a fresh state takes all of them)
    fresh = _to_dev([(np.full_like(v, 0x8000), np.full_like(i, -1)) for v, i in per_rank[0]])
    N.actmax_merge_packed(fresh, gathered, skip_rank=-1)
    for (v, i), (wv, wi) in zip(fresh, want):
        assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(v.view(torch.int16).cpu().numpy().view(np.uint16), wv)


def _native_comm_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    sys.path.insert(0, os.path.dirname(os.path.dirname(__file__)))
    from semanticlens_amd import _native as N
    from semanticlens_amd import distributed as sld

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        assert sld.COLLECTIVES == "native"
        comm = sld.native_comm()
        assert isinstance(comm, N.Comm) and comm.info() == (world, rank, rank) and sld.native_comm() is comm
        # raw collectives
        x = torch.arange(6, dtype=torch.float32, device=dev) + 10 * rank
        got = comm.allgather(x)
        assert got.shape == (world, 6) and all(torch.equal(got[r], torch.arange(6, dtype=torch.float32, device=dev) + 10 * r) for r in range(world))
        for dt in (torch.float32, torch.float64, torch.int64):
            t = torch.full((5,), rank + 1, dtype=dt, device=dev)
            assert comm.allreduce(t.clone(), "sum").tolist() == [world * (world + 1) // 2] * 5
            assert comm.allreduce(t.clone(), "max").tolist() == [world] * 5 and comm.allreduce(t.clone(), "min").tolist() == [1] * 5
        # the fused cross-rank merge
        per_rank, want = _random_states(world, (37, 5, 130), 20, seed=7)
        mine = [(v.to(dev), i.to(dev)) for v, i in _to_dev(per_rank[rank])]
        comm.actmax_allgather_merge(mine)
        ref = want if world > 1 else per_rank[0]  # one rank: nothing to merge, the state keeps its bits
        for (v, i), (wv, wi) in zip(mine, ref):
                assert np.array_equal(v.view(torch.int16).cpu().numpy().view(np.uint16), wv), rank
                assert np.array_equal(i.cpu().numpy(), wi), rank
        # the whole sharded build and the analysis stage through the library's communicator
        cv, fm = _build(device=dev)
        db = sld.compute_concept_db_sharded(cv, fm, batch_size=8)
        rows = torch.arange(35, dtype=torch.float32, device=dev).reshape(7, 5)
        s, e = sld.shard_range(7, rank, world)
        assert torch.equal(sld.all_gather_rows(rows[s:e].contiguous(), 7), rows)
        np.savez(os.path.join(out_dir, f"nc_w{world}_r{rank}.npz"), **{f"db_{k}": v.cpu().numpy() for k, v in db.items()},
                 **{f"ids_{k}": cv.get_max_reference(k).numpy() for k in db})
        sld.destroy_native_comms()
        assert not sld._COMMS
    finally:
        dist.destroy_process_group()


def _check_native_comm_run(world, tmp_path):
    cv, fm = _build()
    want = cv._compute_concept_db(fm, batch_size=8)
    mp.spawn(_native_comm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"nc_w{world}_r{r}.npz")
        for k in ("0", "2"):
            assert np.array_equal(got[f"ids_{k}"], cv.get_max_reference(k).numpy()), (r, k)
            assert np.array_equal(got[f"db_{k}"], want[k].numpy()), (r, k)


def test_library_owned_rccl_communicator_with_one_rank(tmp_path):
    """RCCL inside the C-ABI library.  A one-rank communicator created from a unique id through ctypes runs
    ncclAllGather / ncclAllReduce and `sl_actmax_allgather_merge` on the 1-GPU box; `distributed.py` picks it up for every
    collective of the sharded build when the process group's backend is nccl."""
    _check_native_comm_run(1, tmp_path)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: data movement between GPUs through the library's communicator")
def test_library_owned_rccl_communicator_across_gpus(tmp_path):
    _check_native_comm_run(min(torch.cuda.device_count(), 8), tmp_path)


def test_torch_collectives_remain_selectable(tmp_path, monkeypatch):
    """`SL_COLLECTIVES=torch`: the same build over torch.distributed's all_gather_into_tensor / all_reduce (one-rank nccl group)."""
    monkeypatch.setenv("SL_COLLECTIVES", "torch")
    cv, fm = _build()
    want = cv._compute_concept_db(fm, batch_size=8)
    mp.spawn(_nccl_single_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "nccl1.npz")
    for k in ("0", "2"):
        assert np.array_equal(got[f"db_{k}"], want[k].numpy()), k


# ---- configs[4] names 8 GPUs: the relevance visualizer sharded (both crp states merged across ranks) -------------------------------
def _relevance_cv(device="cuda:0", cache_dir=None):
    from helpers import FakeVLM, TensorPairDataset, make_int_images
    from semanticlens_amd.component_visualization import RelevanceComponentVisualizer
    from test_gpu_relevance import _IntNet, _small_images

    ds = TensorPairDataset(_small_images(29), name="int29")  # 29: shards of unequal size
    ds_fm = TensorPairDataset(make_int_images(29), name="int29-fm")  # what the foundation model embeds (FakeVLM: 3 x 16 x 16)
    cv = RelevanceComponentVisualizer(_IntNet().to(device), ds, ds_fm, ["relu1", "relu2"], num_samples=4, tie_mode="total", cache_dir=cache_dir,
                                      composite="gradient_x_activation", device=device)
    return cv, FakeVLM().to(device)


def _relevance_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from semanticlens_amd import distributed as sld

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cv, fm = _relevance_cv()
        db = sld.compute_concept_db_sharded(cv, fm, batch_size=4)
        np.savez(os.path.join(out_dir, f"rel_w{world}_r{rank}.npz"), **{f"db_{k}": v.cpu().numpy() for k, v in db.items()},
                 **{f"rel_{k}": cv.get_max_reference(k).numpy() for k in db}, **{f"act_{k}": cv.get_act_max_sample_ids(k).numpy() for k in db},
                 **{f"relv_{k}": cv.actmax_cache.cache[k].activations.view(torch.int16).numpy() for k in db})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_relevance_build_equals_single_process(world, tmp_path):
    """`RelevanceComponentVisualizer` under `distributed.compute_concept_db_sharded`: each rank attributes its shard (global ids), both
    states of every layer — relevance mode and activation mode — are merged across ranks, the concept_db is gathered sharded."""
    sys_path = os.path.dirname(__file__)
    import sys

    sys.path.insert(0, sys_path)
    cv, fm = _relevance_cv()
    want = cv._compute_concept_db(fm, batch_size=4)
    mp.spawn(_relevance_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"rel_w{world}_r{r}.npz")
        for k in ("relu1", "relu2"):
            assert np.array_equal(got[f"rel_{k}"], cv.get_max_reference(k).numpy()), (world, r, k)
            assert np.array_equal(got[f"act_{k}"], cv.get_act_max_sample_ids(k).numpy()), (world, r, k)
            assert np.array_equal(got[f"relv_{k}"], cv.actmax_cache.cache[k].activations.view(torch.int16).numpy()), (world, r, k)
            assert np.array_equal(got[f"db_{k}"], want[k].numpy()), (world, r, k)
