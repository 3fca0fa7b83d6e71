"""world_size-2 gloo test (CPU) of the sharded-collect plumbing in semanticlens_amd/distributed.py:
shard ranges, state packing, the single fused all-gather, which ranks get merged.  The K4 merge kernel itself
needs a GPU (tests/test_gpu_parity.py::test_total_mode_is_batch_invariant_and_shard_mergeable); here the
per-rank states come from the oracle and ActMax's two device touch-points are replaced by host stand-ins
that use the oracle as the merge — checker code standing in for the kernel, in a test only."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, acts, k, out_dir):
    import oracle
    from semanticlens_amd import distributed as sld
    from semanticlens_amd.component_visualization import aggregators
    from semanticlens_amd.component_visualization.activation_caching import ActMax, ActMaxCache

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N_, C = acts["a"].shape
        cache = ActMaxCache(["a", "b"], aggregators.aggregate_conv_max, k, tie_mode="total")
        start, stop = sld.shard_range(N_, rank, world)
        for name in ("a", "b"):
            o = oracle.ActMaxOracle(k, acts[name].shape[1], oracle.MODE_TOTAL)
            for s in range(start, stop, 16):
                e = min(stop, s + 16)
                o.update(acts[name][s:e], np.arange(s, e))
            if start == stop:  # empty shard: this rank never saw a batch, its ActMax does not know its width
                cache.cache[name] = ActMax(k, tie_mode="total")
                continue
            am = ActMax(k, acts[name].shape[1], tie_mode="total")
            am.activations = torch.from_numpy(o.vals.view(np.int16)).view(torch.bfloat16)
            am.sample_ids = torch.from_numpy(o.ids)
            cache.cache[name] = am

        def host_state(self, device=None):
            return self.activations, self.sample_ids

        def host_merge(self, other_vals, other_ids):
            o = oracle.ActMaxOracle(self.n_collect, self.n_latents, oracle.MODE_TOTAL)
            o.vals[:] = self.activations.view(torch.int16).numpy().view(np.uint16)
            o.ids[:] = self.sample_ids.numpy()
            o.merge_states(other_vals.contiguous().view(torch.int16).numpy().view(np.uint16), other_ids.numpy())
            self.activations = torch.from_numpy(o.vals.view(np.int16)).view(torch.bfloat16)
            self.sample_ids = torch.from_numpy(o.ids)

        ActMax.device_state = host_state
        ActMax.merge_states = host_merge
        sld.merge_actmax_cache(cache)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
                 **{f"v_{n}": cache.cache[n].activations.view(torch.int16).numpy() for n in ("a", "b")},
                 **{f"i_{n}": cache.cache[n].sample_ids.numpy() for n in ("a", "b")})
    finally:
        dist.destroy_process_group()


def test_sharded_merge_two_ranks_gloo(tmp_path):
    import oracle

    rng = np.random.RandomState(0)
    acts = {"a": (rng.randint(0, 40, size=(150, 12)) / 8.0).astype(np.float32),  # tie-heavy
            "b": np.maximum(rng.randn(150, 5), 0).astype(np.float32)}
    k = 9
    mp.spawn(_worker, args=(2, _free_port(), acts, k, str(tmp_path)), nprocs=2, join=True)
    for name in ("a", "b"):
        ref = oracle.ActMaxOracle(k, acts[name].shape[1], oracle.MODE_TOTAL)
        ref.update(acts[name], np.arange(150))
        for r in (0, 1):
            got = np.load(tmp_path / f"rank{r}.npz")
            assert np.array_equal(got[f"v_{name}"].view(np.uint16), ref.vals), (name, r)
            assert np.array_equal(got[f"i_{name}"], ref.ids), (name, r)


def test_sharded_merge_eight_ranks_gloo(tmp_path):
    """BASELINE configs[2]'s rank count: 8 ranks, 150 samples in shards of 19 (the last one 17), ONE packed all-gather, every rank
    ends with the single-process top-k (tie-heavy values: the total order must not depend on which rank held a sample)."""
    import oracle

    rng = np.random.RandomState(8)
    acts = {"a": (rng.randint(0, 24, size=(150, 10)) / 8.0).astype(np.float32), "b": np.maximum(rng.randn(150, 7), 0).astype(np.float32)}
    k = 11
    mp.spawn(_worker, args=(8, _free_port(), acts, k, str(tmp_path)), nprocs=8, join=True)
    for name in ("a", "b"):
        ref = oracle.ActMaxOracle(k, acts[name].shape[1], oracle.MODE_TOTAL)
        ref.update(acts[name], np.arange(150))
        for r in range(8):
            got = np.load(tmp_path / f"rank{r}.npz")
            assert np.array_equal(got[f"v_{name}"].view(np.uint16), ref.vals), (name, r)
            assert np.array_equal(got[f"i_{name}"], ref.ids), (name, r)


def test_sharded_merge_with_an_empty_shard_gloo(tmp_path):
    """N=2 samples over 3 ranks: the last rank's shard is empty — it must still take part in the all-gather (with
    sentinel states of the agreed width) and end with the global top-k."""
    import oracle

    rng = np.random.RandomState(1)
    acts = {"a": rng.rand(2, 6).astype(np.float32), "b": rng.rand(2, 3).astype(np.float32)}
    k = 4
    mp.spawn(_worker, args=(3, _free_port(), acts, k, str(tmp_path)), nprocs=3, join=True)
    for name in ("a", "b"):
        ref = oracle.ActMaxOracle(k, acts[name].shape[1], oracle.MODE_TOTAL)
        ref.update(acts[name], np.arange(2))
        for r in range(3):
            got = np.load(tmp_path / f"rank{r}.npz")
            assert np.array_equal(got[f"v_{name}"].view(np.uint16), ref.vals), (name, r)
            assert np.array_equal(got[f"i_{name}"], ref.ids), (name, r)


# ---- analysis stage (text_probing_sharded / eval_sharded): row sharding + one all-gather of results ------------------
def _analysis_worker(rank, world, port, out_dir):
    import oracle
    from helpers import FakeVLM
    from semanticlens_amd import _native as N
    from semanticlens_amd import distributed as sld
    from semanticlens_amd import lens as sl_lens

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # uneven rows, an empty shard on the last rank (n_total=4, world=3 -> blocks of 2)
        full = torch.arange(4 * 3, dtype=torch.float32).reshape(4, 3)
        lo, hi = sld.shard_range(4, rank, world)
        got = sld.all_gather_rows(full[lo:hi], 4)
        assert torch.equal(got, full), (rank, got)
        assert sld.all_gather_rows(full[:0], 0).shape == (0, 3)

        # host stand-ins for the two kernels on this path (checker code in a test only)
        N.template_mean = lambda E, E0, Q: torch.from_numpy(oracle.template_mean(E.numpy(), E0.numpy(), Q))
        sl_lens._probe = lambda q, db: {k: torch.from_numpy(oracle.similarity(q.numpy(), v.numpy())) for k, v in db.items()}
        fm = FakeVLM(dim=16)
        rng = np.random.RandomState(3)
        db = {"l1": torch.from_numpy(rng.randn(7, 16).astype(np.float32)), "l2": torch.from_numpy(rng.randn(5, 16).astype(np.float32))}
        queries = ["cat", "dog", "striped zebra", "a", "wheel"]
        templates = ["a photo of a {}", "an image of {}", "{}"]
        for tpl in (None, templates):
            out = sld.text_probing_sharded(fm, queries, db, templates=tpl, batch_size=4)
            np.savez(os.path.join(out_dir, f"probe_{'tpl' if tpl else 'plain'}_rank{rank}.npz"), **{k: v.numpy() for k, v in out.items()})

        # resident query embeddings (bench.py's N > 1 text_probing leg): rows sharded, gathered or kept per rank
        emb = torch.from_numpy(np.random.RandomState(8).randn(6, 16).astype(np.float32))  # (6 queries: no layer has 6 components — a
        lo, hi = sld.shard_range(6, rank, world)                                          #  layer with C == Q takes the reference's shape quirk and stays whole)
        whole, mine = sld.probe_sharded(emb, db), sld.probe_sharded(emb, db, gather=False)
        for k, v in db.items():
            want = oracle.similarity(emb.numpy(), v.numpy())
            assert np.array_equal(whole[k].numpy(), want) and np.array_equal(mine[k].numpy(), want[lo:hi]), (rank, k)

        V = torch.from_numpy(rng.randn(5, 6, 16).astype(np.float32))
        score = lambda v: torch.from_numpy(oracle.clarity(v.numpy()))
        sc = sld.eval_sharded(score, {"x": V})["x"]
        np.save(os.path.join(out_dir, f"clarity_rank{rank}.npy"), sc.numpy())
        # several layers: ONE all-gather for all of them (blocks of ceil(C_l / R) rows per layer); "tiny" leaves ranks 1 and 2 empty
        layers = {"wide": torch.from_numpy(rng.randn(8, 6, 16).astype(np.float32)), "tiny": torch.from_numpy(rng.randn(1, 6, 16).astype(np.float32)),
                  "odd": torch.from_numpy(rng.randn(5, 6, 16).astype(np.float32))}
        calls = []
        real = sld.all_gather_rows
        sld.all_gather_rows = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
        try:
            multi = sld.eval_sharded(score, layers)
        finally:
            sld.all_gather_rows = real
        assert len(calls) == 1 and list(multi) == list(layers)
        np.savez(os.path.join(out_dir, f"clarity_multi_rank{rank}.npz"), **{k: v.numpy() for k, v in multi.items()},
                 **{f"in_{k}": v.numpy() for k, v in layers.items()})
    finally:
        dist.destroy_process_group()


def test_sharded_probing_and_scores_three_ranks_gloo(tmp_path):
    import oracle
    from helpers import FakeVLM
    from semanticlens_amd import lens as sl_lens

    world = 3
    mp.spawn(_analysis_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    # single-process expectation with the same stand-ins
    fm = FakeVLM(dim=16)
    rng = np.random.RandomState(3)
    db = {"l1": rng.randn(7, 16).astype(np.float32), "l2": rng.randn(5, 16).astype(np.float32)}
    queries = ["cat", "dog", "striped zebra", "a", "wheel"]
    templates = ["a photo of a {}", "an image of {}", "{}"]
    enc = lambda texts: fm.encode_text(fm.tokenize(texts)).numpy()
    plain = enc(queries)
    templated = oracle.template_mean(enc([t.format(q) for t in templates for q in queries]), enc([t.format("") for t in templates]), len(queries))
    for tag, emb in (("plain", plain), ("tpl", templated)):
        for r in range(world):
            got = np.load(tmp_path / f"probe_{tag}_rank{r}.npz")
            for k, v in db.items():
                assert np.array_equal(got[k], oracle.similarity(emb, v)), (tag, r, k)
    V = rng.randn(5, 6, 16).astype(np.float32)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"clarity_rank{r}.npy"), oracle.clarity(V))
        got = np.load(tmp_path / f"clarity_multi_rank{r}.npz")
        for k in ("wide", "tiny", "odd"):
            assert got[k].shape == (got[f"in_{k}"].shape[0],) and np.array_equal(got[k], oracle.clarity(got[f"in_{k}"])), (r, k)
