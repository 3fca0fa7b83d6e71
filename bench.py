"""bench.py — concept-DB build throughput (BASELINE.json configs[1]) + text_probing, on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3        # launches its own 8 ranks (one per GPU, RCCL), prints ONE line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over `--batches-per-step` (10) batches of B = 256 synthetic images already resident
in HBM — 2 560 images, so that the driver's `--steps 20` covers BASELINE configs[1]'s 50 000 images (51 200).  Per batch:
ResNet-50 forward under the collect hooks (K1 reduce + K3 top-k merge for layer2/3/4) and the CLIP
ViT-B/32 image encode of the same batch into the device-resident embedding table.  After the K timed
steps the job is finished inside the timed region: pending merges are flushed, (N>1: per-rank
top-k states are all-gathered over RCCL and merged, K4) and the concept_db of every layer is
gathered (K5).  `value` = images of all ranks / max-over-ranks wall time.

One JSON line is printed by rank 0; see DESIGN.md "Measurement" for every field.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import synth  # noqa: E402
from semanticlens_amd import Lens  # noqa: E402
from semanticlens_amd import _native as N  # noqa: E402
from semanticlens_amd import distributed as sld  # noqa: E402
from semanticlens_amd.component_visualization import ActivationComponentVisualizer, aggregators  # noqa: E402

LAYERS = ["layer2", "layer3", "layer4"]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)
HBM_COPY_CEILING_GBPS = 6290.0  # MI355X_MICROARCH.md: the measured copy ceiling of this part (6.29 TB/s)


def roofline_hbm(gbps, **more):
    """An HBM roofline object: `frac` against the 8 TB/s spec (the contract's denominator) and, beside it, against the copy ceiling the
    microarchitecture guide measured on this part — north_star's >= 90 % target is judged against both, visibly."""
    out = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS if gbps else None,
           "frac_of_measured_copy_ceiling": gbps / HBM_COPY_CEILING_GBPS if gbps else None,
           "measured_copy_ceiling_GBps": HBM_COPY_CEILING_GBPS}
    traffic = more.pop("traffic", None)
    out["traffic"] = traffic
    out.update(more)
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)  # 20 steps x 10 batches x 256 = 51,200 images (configs[1]: "50k")
    ap.add_argument("--batches-per-step", type=int, default=10, help="batches of --batch images one step walks")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=20)  # num_samples; the reference tutorial's value
    ap.add_argument("--tie-mode", default=None, choices=["total", "aten"],
                    help="default: 'aten' on one GPU (top-k ids in torch.topk's CPU tie order = bit-identical to the reference, "
                         "activation_caching.py:133-141), 'total' on several (value desc, id asc: shard- and batch-invariant)")
    ap.add_argument("--strong-images", type=int, default=1281167,
                    help="after the weak-scaling job, build the concept DB of this many images in TOTAL, sharded over the ranks "
                         "(north_star: 1.28 M-image set, >= 6x from 1 to 8 GPUs) and report it as `strong_scaling`; 0 = skip")
    ap.add_argument("--strong-pool-batches", type=int, default=200, help="distinct resident batches the strong-scaling job cycles through")
    ap.add_argument("--cpu-images", type=int, default=1024, help="bounded sample for the CPU baseline leg (about 50 s at the ~20 images/s of these hosts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probing", action="store_true")
    ap.add_argument("--fm", default="native", choices=["native", "native-f32", "torch"],
                    help="CLIP ViT-B/32 encoder: package kernels (split-bf16 x3 or fp32 MFMA GEMMs) or the torch module")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="run the CLIP embed of each batch on the same HIP stream as forward + collect (default: a second "
                         "stream, +6 %% images/s; K1 then shares HBM with the encoder: in-bench roofline fraction 0.737 vs 0.75)")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (exhaustive MIOpen search)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --steps batches PER RANK (the driver's contract); strong: --images samples in TOTAL, sharded "
                         "over the ranks by distributed.shard_range (north_star: 1.28 M images, >= 6x at 8 GPUs)")
    ap.add_argument("--images", type=int, default=0, help="--scaling strong: total dataset size (default steps x batch)")
    ap.add_argument("--pool-batches", type=int, default=0,
                    help="keep only this many distinct batches resident and cycle through them (ids stay unique); 0 = every "
                         "batch distinct.  For dataset sizes whose uint8 pixels exceed HBM (1.28 M images = 193 GB)")
    ap.add_argument("--leg-steps", type=int, default=0, help="batches of the configs[3] / configs[4] collect legs (default 8 / 4; tests: fewer)")
    ap.add_argument("--no-self-check", action="store_true")
    ap.add_argument("--quick", action="store_true", help="headline + self-check only: none of the extra legs (tests)")
    ap.add_argument("--no-tokens-leg", action="store_true", help="skip the configs[3] leg (ViT-B/16 collect + so400m embed + 10k-prompt probing)")
    ap.add_argument("--no-config4-leg", action="store_true", help="skip the configs[4] leg (ConvNeXt-L collect + relevance visualizer + scores)")
    ap.add_argument("--no-half-leg", action="store_true")
    ap.add_argument("--no-api-leg", action="store_true")
    ap.add_argument("--no-channels-last", action="store_true")
    ap.add_argument("--min-warmup-seconds", type=float, default=1.5,
                    help="after the first --warmup job, repeat it for at least this long and until it runs at a settled speed, at most "
                         "ten times as long (0: exactly --warmup steps)")
    ap.add_argument("--api-images", type=int, default=8192,
                    help="raw 500x375 images of the API-path leg (9.5 GB of host memory at 8192; at 2048 the pipeline's fill and the "
                         "final device-to-host copy of the concept DB cost 12 %% of the run)")
    return ap.parse_args()


class _Len:
    """Length-only dataset stand-in: the timed loop feeds device-resident batches directly."""

    def __init__(self, n, name):
        self.n, self.name = n, name

    def __len__(self):
        return self.n


def make_cv(model, n_total, k, tie_mode, layers=None, agg=None):
    return ActivationComponentVisualizer(
        model, _Len(n_total, f"synthetic-{n_total}"), _Len(n_total, "synthetic-fm"), list(layers or LAYERS), num_samples=k,
        aggregate_fn=agg or aggregators.aggregate_conv_max, cache_dir=None, tie_mode=tie_mode,
    )


OVERLAP = True  # embed on a second HIP stream beside forward + collect (--no-overlap: one stream)


class _Ctx:
    """Who this process is in the job (set once by main): every leg reads it, so that the same leg code runs alone (N = 1) or as one
    of N ranks — per-rank work between two barriers, max-over-ranks wall time, aggregate units / that time."""

    rank, world, sharded, backend, comm, dev = 0, 1, False, "nccl", None, None

    def all_max(self, x: float) -> float:
        """max over ranks of one host double (timing, warm-up decision)"""
        if not self.sharded:
            return x
        if self.comm is not None:
            t = torch.tensor([x], dtype=torch.float64, device=self.dev)
            self.comm.allreduce(t, "max")
            return float(t.item())
        t = torch.tensor([x], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        torch.cuda.synchronize()
        if self.sharded:
            dist.barrier()

    def timed(self, fn):
        """fn() between two barriers (+ device synchronisation on both sides); returns (max-over-ranks seconds, fn's result)"""
        self.barrier()
        t0 = time.perf_counter()
        res = fn()
        self.barrier()
        return self.all_max(time.perf_counter() - t0), res


CTX = _Ctx()


@torch.no_grad()
def run_steps(cv, fm, batches, id_start, n_local, cast=None):
    """The timed inner loop over device-resident uint8 batches (``cast``: dtype of a half-precision probed model)."""
    for name in cv.layer_names:
        cv.actmax_cache.sample_idx_counter[name] = id_start
    embeds, filled = None, 0
    overlap = OVERLAP
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream() if overlap else None

    def model_input(u8):
        x = synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD)
        return x if cast is None else x.to(cast)

    with cv.actmax_cache.hook_context(cv.model):
        for u8 in batches:
            if overlap:  # hot loop 2 (embed) on a second HIP stream beside hot loop 1 (forward + collect)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    embeds, filled = cv.embed_batch(fm, u8, embeds, filled, n_local)
                cv.collect_batch(model_input(u8))
            else:
                cv.collect_batch(model_input(u8))
                embeds, filled = cv.embed_batch(fm, u8, embeds, filled, n_local)
    if overlap:
        main.wait_stream(side)
    return embeds


@torch.no_grad()
def finish_job(cv, embeds, id_start, n_total, sharded):
    """Flush/merge the top-k states and gather the concept_db (device tensors).  ``sharded``: a process group is up —
    all-gather + K4 merge of the per-rank states, sharded K5 + all-reduce (also with ONE rank: SL_BENCH_FORCE_DIST)."""
    if sharded:
        sld.merge_actmax_cache(cv.actmax_cache)
        return {n: sld.gather_concept_db_sharded(embeds, id_start, n_total, cv.get_max_reference(n)) for n in cv.layer_names}
    # one K5 launch over the ids of all layers (what `_compute_concept_db` runs), straight from the device-resident states
    return cv._gather_layers(embeds, {n: cv.actmax_cache.cache[n].device_state()[1] for n in cv.layer_names}, True)


@torch.no_grad()
def self_check(dev, model, fm, args, n=512, B=256, layers=None, agg=None, cast=None):
    """What the timed region computes, checked against the oracle on an n-image prefix (ids 0..n-1, batches of the
    bench's own size — other batch sizes would send MIOpen into a fresh kernel search): the SAME device activations go
    through (i) the product's hooks
    (K1/K2 reduce + K3 merge, the tie mode of the timed run) and (ii) the oracle's aggregate + ActMax restatement on the
    host; top-k values and ids must be bit-equal, and the gathered concept_db (K5) must equal the oracle's gather of
    the device embeddings.  Raises on any difference."""
    import numpy as np

    import oracle  # the checker — never on the product path

    layers = list(layers or LAYERS)
    agg = agg or aggregators.aggregate_conv_max
    tokens = agg._sl_native[0] == "tokens"
    cv = make_cv(model, n, args.k, args.tie_mode, layers, agg)
    mode = oracle.MODE_TOTAL if args.tie_mode == "total" else oracle.MODE_ATEN
    refs, seen = {}, {name: 0 for name in layers}

    def tap(name):
        def fn(m, i, o):  # oracle side, streamed: aggregate on the host, merge into the oracle's state
            act = o.detach().float().cpu().numpy()
            a = oracle.agg_tokens(act, "max") if tokens else oracle.agg_conv(act, "max")
            if name not in refs:
                refs[name] = oracle.ActMaxOracle(args.k, a.shape[1], mode)
            refs[name].update(a, np.arange(seen[name], seen[name] + a.shape[0]))
            seen[name] += a.shape[0]

        return fn

    modules = dict(model.named_modules())
    taps = [modules[name].register_forward_hook(tap(name)) for name in layers]
    batches = [synth.synth_images_u8(torch.arange(s, min(n, s + B), device=dev)) for s in range(0, n, B)]
    try:
        embeds = run_steps(cv, fm, batches, 0, n, cast)
    finally:
        for h in taps:
            h.remove()
    db = finish_job(cv, embeds, 0, n, False)
    torch.cuda.synchronize()
    emb_host = embeds.cpu().numpy()
    for name in layers:
        ref = refs[name]
        am = cv.actmax_cache.cache[name]
        got_v = am.activations.view(torch.int16).numpy().view(np.uint16)
        if not np.array_equal(got_v, ref.vals):
            raise AssertionError(f"self-check: top-k values of {name} differ from the oracle")
        if not np.array_equal(am.sample_ids.numpy(), ref.ids):
            raise AssertionError(f"self-check: top-k ids of {name} differ from the oracle")
        if not np.array_equal(db[name].cpu().numpy(), oracle.gather_rows(emb_host, ref.ids)):
            raise AssertionError(f"self-check: concept_db of {name} differs from the oracle's gather")
    return "ok"


@torch.no_grad()
def reduce_cold_leg(dev, B):
    """K1 alone on COLD inputs at the three layer shapes (kernel-only; inputs rotated through > 1.2 GB so that the 256 MiB
    Infinity Cache cannot serve them; read-once cache policy, as for any input that was not written a moment ago).  The
    headline `roofline` is the in-pipeline number; this is what the same kernel does when its input comes from HBM."""
    out = {}
    N.set_reduce_policy(0, 0)
    try:
        for name, (C, H) in zip(LAYERS, ((512, 28), (1024, 14), (2048, 7))):
            nbytes = B * C * H * H * 4
            copies = max(2, (1200 << 20) // nbytes + 1)
            xs = [torch.rand(B, C, H, H, device=dev) for _ in range(copies)]
            cand = torch.empty((B, C), dtype=torch.bfloat16, device=dev)
            for x in xs:
                N.reduce_conv(x, N.SL_CONV_MAX, cand, None)
            torch.cuda.synchronize()
            N.prof_enable(True)
            N.prof_reset()
            for _ in range(3):
                for x in xs:
                    N.reduce_conv(x, N.SL_CONV_MAX, cand, None)
            torch.cuda.synchronize()
            ms, launches, nb = N.prof_read(N.SL_PROF_REDUCE)
            N.prof_enable(False)
            out[name] = {"GB/s": nb / ms / 1e6, "frac": nb / ms / 1e6 / HBM_PEAK_GBPS, "avg_launch_us": ms / launches * 1e3,
                         "bytes_per_launch": nbytes}
            del xs
    finally:
        N.set_reduce_policy(None, None)
    tot_b = sum(v["bytes_per_launch"] for v in out.values())
    tot_t = sum(v["bytes_per_launch"] / v["GB/s"] for v in out.values())
    out["all_layers"] = {"GB/s": tot_b / tot_t, "frac": tot_b / tot_t / HBM_PEAK_GBPS}
    return out


class _Rows(torch.utils.data.Dataset):
    """In-memory dataset over the first axis of a host tensor (what a user's Dataset hands the reference's loops)."""

    def __init__(self, x, name, with_label):
        self.x, self.name, self.with_label = x, name, with_label

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return (self.x[i], 0) if self.with_label else self.x[i]


@torch.no_grad()
def api_path_leg(dev, model, fm_base, args):
    """The path a user runs (activation_based.py:341-433 through `Lens.compute_concept_db`): two host-resident
    `Dataset`s — normalised 224x224 fp32 samples for the probed model, RAW 500x375 uint8 images for the foundation
    model — walked by torch DataLoaders, `NativeClip(preprocess=DevicePreprocess(224))` doing resize / crop / normalise
    on the device (K12), `single_pass=True`, default tie mode ("aten": bit-identical to the reference).  PCIe- and
    DataLoader-inclusive; never the headline `value`."""
    from semanticlens_amd.foundation_models import DevicePreprocess
    from semanticlens_amd.foundation_models.native_clip import NativeClip

    n, B, h, w = args.api_images, args.batch, 375, 500
    g = torch.Generator(device=dev).manual_seed(5)
    raw = torch.empty((n, h, w, 3), dtype=torch.uint8)
    norm = torch.empty((n, 3, 224, 224), dtype=torch.float32)
    to_model = DevicePreprocess(224, synth.IMAGENET_MEAN, synth.IMAGENET_STD, device=dev)
    for s in range(0, n, B):
        e = min(n, s + B)
        # smooth low-frequency content + noise, so the resample is not a no-op
        base = torch.rand((e - s, 12, 16, 3), device=dev, generator=g).permute(0, 3, 1, 2)
        img = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)
        img = (img * 200 + 55 * torch.rand((e - s, 3, h, w), device=dev, generator=g)).clamp_(0, 255).to(torch.uint8)
        u8 = img.permute(0, 2, 3, 1).contiguous()
        raw[s:e] = u8.cpu()
        norm[s:e] = to_model(u8).cpu()
    fm = NativeClip(fm_base, gemm="bf16x3", preprocess=DevicePreprocess(224, synth.CLIP_MEAN, synth.CLIP_STD))
    lens = Lens(fm, device=dev)

    def build(n_use, single_pass=True):
        cv = ActivationComponentVisualizer(
            model, _Rows(norm[:n_use], f"api-{n_use}", True), _Rows(raw[:n_use].numpy(), "api-fm", False), LAYERS,
            num_samples=args.k, aggregate_fn=aggregators.aggregate_conv_max, cache_dir=None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        db = lens.compute_concept_db(cv, batch_size=B, single_pass=single_pass)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, db

    # torch's CPU collate (`torch.stack` of 256 samples = 154 MB per batch) takes 0.8 s with the box's default of 100+ intra-op
    # threads and ~0.05 s with 16: host configuration of the DataLoader side, set like a user would
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(16, threads_before))
    try:
        build(min(n, 2 * B))  # warm-up
        dt, db = build(n)
        build(min(n, 2 * B), single_pass=False)
        dt2, db2 = build(n, single_pass=False)  # the reference's own call: two sequential passes over the data
        same = all(torch.equal(db[k_], db2[k_]) for k_ in db)
        # below B = 256 the probed model's own forward is not run-to-run deterministic on this stack (MIOpen picks convolution
        # algorithms with atomics for small batches: tools/api_small_batch.py), so two walks may keep different top-k samples
        assert same or B < 256
    finally:
        torch.set_num_threads(threads_before)
    assert all(v.shape == (c, args.k, 512) for v, c in zip(db.values(), (512, 1024, 2048)))
    return {"api_path_images_per_s": n / dt, "images": n, "seconds": dt, "two_pass_images_per_s": n / dt2, "single_pass_equals_two_pass": same,
            "workload": f"Lens.compute_concept_db(cv, batch_size={B}, single_pass=True): host Datasets ({n} normalised "
                        f"224x224 fp32 samples + raw {w}x{h} uint8 images), DataLoader num_workers=0 walked by the background prefetch threads (pinned staging, uploads ahead of the device), 16 host threads, device preprocessing "
                        "(K12), tie_mode='aten', concept_db returned on the host"}


def cpu_baseline(args, model_cpu, fm_cpu):
    """The CPU path (oracle = port of the reference's arithmetic) on a bounded sample of the same
    workload, on this box's host cores: torch-CPU forward under hooks -> oracle aggregate ->
    oracle ActMax.update (torch.topk tie order) ; CLIP encode on CPU ; oracle gather."""
    import numpy as np

    import oracle  # checker / baseline only — never on the product path

    B = 64
    # pick the torch thread count that is fastest on this box (containers often expose more logical CPUs than their quota
    # sustains: on the MI355X boxes 8-32 threads beat 256 by 20-70x) for ONE BATCH'S COMPUTE of the job — ResNet-50 forward + CLIP
    # encode of 32 images after an 8-image call of both (the forward alone picked 2 threads on one box of round 6: 32 images/s for
    # the forward, 10 for the job).  Counts past the first one that falls below half of the best are not probed (a 64-image forward
    # with one thread per logical CPU of a 256-CPU box costs minutes); all cores get a 4-image probe in that case.
    probe_u8 = synth.synth_images_u8(torch.arange(10**7, 10**7 + 32))
    try:
        all_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        all_cores = os.cpu_count() or torch.get_num_threads()
    best_t, best_rate = torch.get_num_threads(), 0.0
    cands = sorted({t for t in (2, 4, 8, 16, 32, 64, 128, all_cores) if t <= all_cores})
    thread_probe = {}

    def batch_compute(u8):
        model_cpu(synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD))
        fm_cpu.encode_image(fm_cpu.preprocess(u8))

    stop = False
    with torch.no_grad():
        for t in cands:
            if stop and t != all_cores:
                continue
            torch.set_num_threads(t)
            if stop:  # all cores, known to be far off (18-36 s PER IMAGE on the boxes of round 6): one image, no warm-up call
                t0 = time.perf_counter()
                batch_compute(probe_u8[:1])
                thread_probe[t] = 1 / (time.perf_counter() - t0)
                continue
            batch_compute(probe_u8[:8])
            t0 = time.perf_counter()
            batch_compute(probe_u8)
            thread_probe[t] = 32 / (time.perf_counter() - t0)
            if thread_probe[t] > best_rate:
                best_t, best_rate = t, thread_probe[t]
            stop = thread_probe[t] < 0.5 * best_rate
    bytes_per_img = 2809856  # ResNet-50 layer2+3+4 fp32 activations per image

    def job(n, threads, warm=True):
        """the whole CPU job over n images with `threads` torch / OpenMP threads: (images/s, seconds, seconds inside the collect)"""
        torch.set_num_threads(threads)
        oracle.set_threads(threads)
        states, grabbed, agg_s = {}, {n_: 0 for n_ in LAYERS}, [0.0]

        def hook(name):
            def fn(m, i, o):
                t = time.perf_counter()
                a = oracle.agg_conv(o.detach().numpy(), "max")
                if name not in states:
                    states[name] = oracle.ActMaxOracle(args.k, a.shape[1], oracle.MODE_ATEN)
                states[name].update(a, np.arange(grabbed[name], grabbed[name] + a.shape[0]))
                grabbed[name] += a.shape[0]
                agg_s[0] += time.perf_counter() - t

            return fn

        handles = [getattr(model_cpu, n_).register_forward_hook(hook(n_)) for n_ in LAYERS]
        try:
            if warm:
                with torch.no_grad():  # one untimed batch: thread pools, oneDNN primitive caches
                    u8 = synth.synth_images_u8(torch.arange(10**7, 10**7 + B))
                    model_cpu(synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD))
                    fm_cpu.encode_image(fm_cpu.preprocess(u8))
                states.clear()
                for n_ in LAYERS:
                    grabbed[n_] = 0
                agg_s[0] = 0.0
            embeds = []
            t0 = time.perf_counter()
            with torch.no_grad():
                for s in range(0, n, B):
                    u8 = synth.synth_images_u8(torch.arange(s, s + B))
                    model_cpu(synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD))
                    embeds.append(fm_cpu.encode_image(fm_cpu.preprocess(u8)).numpy())
            emb = np.concatenate(embeds)
            for n_ in LAYERS:
                oracle.gather_rows(emb, states[n_].ids)
            dt = time.perf_counter() - t0
        finally:
            for h in handles:
                h.remove()
        return n / dt, dt, agg_s[0]

    def job_b(b_, threads_):
        nonlocal B
        saved = B
        B = b_
        try:
            return job(b_, threads_, warm=False)
        finally:
            B = saved

    threads = best_t
    n = max(B, (args.cpu_images // B) * B)
    rate, dt, agg_s = job(n, threads)
    # every core of the box, the same job on ONE batch (SURVEY §8d planned "all host cores"; on these boxes one thread per logical CPU is
    # an order of magnitude slower than the best count, so its sample is kept to one batch; pools are warm from the thread probe)
    B_all = 8  # (64 images took 80 s with 256 threads on the round's first box, 16 images 73 s on its third)
    all_sample = None
    if all_cores == threads:
        rate_all, dt_all = rate, dt
    elif stop:
        # one thread per logical CPU already measured far below half of the best count by the probe: its probe figure (forward +
        # encode of one image) stands for it — the whole job there would add minutes to the line for a number nobody can use
        rate_all, dt_all = thread_probe[all_cores], 1 / thread_probe[all_cores]
        all_sample = f"1 image, ResNet-50 forward + CLIP encode only (the thread probe), {dt_all:.1f} s"
    else:
        rate_all, dt_all, _ = job_b(B_all, all_cores)
        torch.set_num_threads(threads)
        oracle.set_threads(threads)
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = None
    return {
        # `cores` = the threads the run USED (the fastest of 2/4/8/16/32/64/128/all on this box for the torch-CPU forward);
        # `host_cores` = what the box has (logical CPUs / CPUs this process may run on).  Best count and all cores side by side:
        "value": rate, "unit": "images/s", "cores": threads, "threads": threads,
        "all_cores": {"value": rate_all, "unit": "images/s", "cores": all_cores,
                      "sample": all_sample or f"{B_all if all_cores != threads else n} images (one batch), the same job, {dt_all:.1f} s",
                      "probe_images_per_s": thread_probe.get(all_cores)},
        "host_cores": os.cpu_count(), "host_cores_affinity": affinity, "kind": "port",
        "sample": f"{n} synthetic images (batch {B}), same models/layers/k, torch-CPU forward + oracle collect "
                  f"(ATen tie order) + CPU CLIP encode + gather; {dt:.1f} s",
        "collect_only_GBps": n * bytes_per_img / agg_s / 1e9,
        "collect_only_seconds": agg_s,
        # one batch's compute (ResNet-50 forward + CLIP encode, 32 images), images/s per thread count tried: how `cores` was chosen
        "thread_probe_images_per_s": {str(k_): v for k_, v in thread_probe.items()},
    }


@torch.no_grad()
def probing_end_to_end(fm, dev):
    """Lens.text_probing on 10,000 synthetic prompts: tokenise + text tower + template-free probe against 12 x 768
    concept embeddings in the tower's own joint space (D=512 for ViT-B/32)."""
    words = ["zebra", "stripe", "wheel", "sky", "grass", "dog", "cat", "red", "round", "metal", "wood", "water", "face", "text"]
    prompts = [f"a photo of a {words[i % 14]} {words[(i // 14) % 14]} {i}" for i in range(10000)]
    g = torch.Generator(device=dev).manual_seed(3)
    db = {f"block{i}": torch.randn(768, 512, device=dev, generator=g) for i in range(12)}
    lens = Lens(fm, device=dev)
    if CTX.sharded:
        def probe(p):
            return sld.text_probing_sharded(fm, p, db, batch_size=1024)
    else:
        def probe(p):
            return lens.text_probing(p, db, batch_size=1024)
    probe(prompts[: 1024 * CTX.world])  # warm-up
    wall, out = CTX.timed(lambda: probe(prompts))
    assert all(v.shape == (10000, 768) for v in out.values())
    return {"queries_per_s": 10000 / wall, "wall_ms": wall * 1e3, "n_gpus": CTX.world,
            "workload": "10,000 prompts -> tokenizer -> text tower (12 x 512, ctx 77, batches of 1024) -> probe vs 12 x 768 x 512"
                        + ("; prompts and query rows sharded over the ranks (distributed.text_probing_sharded)" if CTX.sharded else "")}


def probing_leg(dev):
    """text_probing at BASELINE configs[3] shapes: Q=10,000 query embeddings (D=1152, SigLIP-so400m width) against
    12 layers x 768 components, through `_probe` (lens.py:206-214).  The cosine GEMM (K6) is timed per dispatch
    through sl_prof.  Measured in both arithmetic modes; the default (split-bf16 x3) is the headline.
    N ranks: the query rows are sharded (distributed.probe_sharded: rank r probes rows shard_range(Q, r, N) against the replicated
    DB); `value` = all similarities / max-over-ranks wall INCLUDING the all-gather that leaves the full (Q, C) result of every layer
    on every rank — what `text_probing_sharded` returns; the compute-only rate (each rank keeps its rows) is beside it."""
    import numpy as np

    import oracle  # checker only

    g = torch.Generator(device=dev).manual_seed(2)
    Q, D, C, L = 10000, 1152, 768, 12
    q = torch.randn(Q, D, device=dev, generator=g)
    db = {f"block{i}": torch.randn(C, D, device=dev, generator=g) for i in range(L)}
    from semanticlens_amd.lens import _probe

    sharded, world = CTX.sharded, CTX.world
    reps = 8  # probe calls timed back to back (normalise + split + GEMM each) after 0.4 s of untimed ones:
    # the first few hundred ms of this load run 4-6 % below the steady state (403 -> 440 TFLOP/s over repetitions of the leg)

    def call(gather=True):
        return sld.probe_sharded(q, db, gather=gather) if sharded else _probe(q, db)

    def run(mode, gather=True):
        N.set_gemm_mode(mode)
        t_w = time.perf_counter()
        more = 1.0
        while more:  # warm-up: the part needs a few hundred ms of this load to settle (every rank runs the same count)
            for _ in range(reps):
                call(gather)
            torch.cuda.synchronize()
            more = CTX.all_max(1.0 if time.perf_counter() - t_w < 0.4 else 0.0)
        N.prof_enable(True)
        N.prof_reset()
        out = [None]

        def body():
            for _ in range(reps):
                out[0] = call(gather)

        wall, _ = CTX.timed(body)
        wall /= reps
        ms, launches, flops = N.prof_read(N.SL_PROF_GEMM)
        N.prof_enable(False)
        return wall, ms, launches, flops, out[0]

    wall3, ms3, n3, fl3, out3 = run("bf16x3")
    wall1, ms1, n1, fl1, out1 = run("f32")
    assert all(v.shape == (Q, C) for v in out3.values())
    max_diff = max((out3[k] - out1[k]).abs().max().item() for k in out3)
    sims = Q * C * L
    res = {
        "metric": "Msimilarities/sec text_probing", "value": sims / wall3 / 1e6, "unit": "Msim/s", "n_gpus": world,
        "workload": f"Q={Q} x {L} layers x C={C}, D={D} (configs[3] shapes), query embeddings resident; mean of {reps} calls"
                    + (f"; query rows sharded over {world} ranks, similarity rows all-gathered (every rank ends with all {L} (Q, C) results)"
                       if sharded else ""),
        "wall_ms": wall3 * 1e3,
        # split-bf16 x3: three bf16 MFMAs per product.  `achieved` counts ALGORITHMIC flops (2*Q*C*D); the peak it is
        # priced against is the dense bf16 MFMA peak divided by the 3 products (2500 / 3); 3x achieved is what the
        # matrix cores actually issue.
        "roofline": {"bound": "mfma", "achieved": fl3 / ms3 / 1e9, "peak": MFMA_BF16_PEAK_TFLOPS / 3, "unit": "TFLOP/s",
                     "frac": (fl3 / ms3 / 1e9) / (MFMA_BF16_PEAK_TFLOPS / 3),
                     "kernel": "gemm3_nt_8phase (split-bf16 x3 on v_mfma_f32_32x32x16_bf16, 256x256 tiles, two staggered wave groups, 16-KB LDS-DMA pieces six ahead, all 12 layers in one launch; fp32-class accuracy)"
                               + (f"; rank 0's launches: {-(-Q // world)} query rows each" if sharded else ""),
                     "mfma_flops_issued_TFLOPs": 3 * fl3 / ms3 / 1e9,
                     "ratio_to_fp32_mfma_peak": (fl3 / ms3 / 1e9) / MFMA_F32_PEAK_TFLOPS,
                     "launches": n3, "avg_ms": ms3 / max(n3, 1)},
        # the same probe with the exact-fp32 kernel (SL_GEMM_MODE=f32)
        "fp32_mfma_mode": {"value": sims / wall1 / 1e6, "unit": "Msim/s", "wall_ms": wall1 * 1e3,
                           "roofline": {"bound": "mfma", "achieved": fl1 / ms1 / 1e9, "peak": MFMA_F32_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": (fl1 / ms1 / 1e9) / MFMA_F32_PEAK_TFLOPS,
                                        "kernel": "gemm_nt_8phase<f32> (v_mfma_f32_32x32x2_f32, 256x256 tiles, all 12 layers gathered into one launch)", "launches": n1,
                                        "avg_ms": ms1 / max(n1, 1)}},
        "max_abs_diff_between_modes": max_diff,
    }
    if sharded:
        wall_c, _, _, _, part = run("bf16x3", gather=False)
        res["compute_only"] = {"value": sims / wall_c / 1e6, "unit": "Msim/s", "wall_ms": wall_c * 1e3,
                               "note": "every rank keeps the similarity rows of its own queries (no all-gather of the 369 MB result)"}
        # parity of the sharded probe (rank 0): every bit of the single-process call, and the oracle's float64 cosine on 64 query rows
        N.set_gemm_mode("bf16x3")
        whole = _probe(q, db)
        keys = list(db)
        want = oracle.similarity(q[:64].cpu().numpy(), db[keys[-1]].cpu().numpy())
        res["sharded_check"] = {
            "equals_single_process_bitwise": all(torch.equal(out3[k], whole[k]) for k in keys),
            "max_abs_diff_vs_oracle_64_queries": float(np.abs(out3[keys[-1]][:64].cpu().numpy() - want).max())}
        del whole, part
    N.set_gemm_mode(None)
    return res


def policy_report(cv):
    """Per hooked layer: which cache policy its reduce settled on (N.ReducePolicyTuner: 'default' = inputs below 256 MiB and the last
    240 MiB of larger ones with the default policy; 'nt>=96MiB,tail80MiB' = what outputs of residual adds want)."""
    names = {None: "not tuned (input below 96 MiB, or policy set by the caller)", 0: "default", 1: "nt>=96MiB,tail80MiB",
             2: "nt>=96MiB,tail128MiB"}
    out = {}
    for name in cv.layer_names:
        t = cv.actmax_cache.cache[name]._policy_tuner
        out[name] = names.get(getattr(t, "choice", None), "measuring") if t is not None and getattr(t, "_key", None) is not None else names[None]
    return out


def collect_leg(dev, fm, args, model, layers, agg, kernel, workload, steps, B, cast=None, check_n=None, overlap=None, keep_db=None,
                traffic_key=None):
    """A short run of the same step on another probed model / aggregator / activation dtype, so that the reduce kernel that
    configuration selects gets its own driver-measured roofline object (algorithmic bytes / per-dispatch HIP-event time, as
    for the headline) and its own oracle self-check.  ``overlap``: embed on a second stream (None = as the headline)."""
    global OVERLAP
    saved = OVERLAP
    if overlap is not None:
        OVERLAP = bool(overlap)
    try:
        return _collect_leg(dev, fm, args, model, layers, agg, kernel, workload, steps, B, cast, check_n, keep_db, traffic_key)
    finally:
        OVERLAP = saved


def _collect_leg(dev, fm, args, model, layers, agg, kernel, workload, steps, B, cast, check_n, keep_db=None, traffic_key=None):
    rank, world, sharded = CTX.rank, CTX.world, CTX.sharded
    # MIOpen / hipBLASLt pick their kernels, and each hooked layer's reduce settles its cache policy (N.ReducePolicyTuner: 8 launches)
    warm = [synth.synth_images_u8(torch.arange(10**7 + (i % 2) * B, 10**7 + (i % 2 + 1) * B, device=dev)) for i in range(2)] * 7
    cv_w = make_cv(model, world * 14 * B, args.k, args.tie_mode, layers, agg)
    finish_job(cv_w, run_steps(cv_w, fm, warm, rank * 14 * B, 14 * B, cast), rank * 14 * B, world * 14 * B, sharded)
    # N ranks: rank r walks the contiguous ids [r * n, (r + 1) * n) of an (N * n)-image set; the job ends with the cross-rank merge
    # (packed all-gather + K4) and the sharded gather, like the headline
    n = steps * B
    n_total, id0 = world * n, rank * n
    batches = [synth.synth_images_u8(torch.arange(id0 + s * B, id0 + (s + 1) * B, device=dev)) for s in range(steps)]
    cv = make_cv(model, n_total, args.k, args.tie_mode, layers, agg)
    N.prof_enable(True)
    N.prof_reset()
    CTX.barrier()
    t0 = time.perf_counter()
    emb = run_steps(cv, fm, batches, id0, n, cast)
    table_bytes = emb.numel() * emb.element_size()
    torch.cuda.synchronize()
    enc_g = N.prof_read(N.SL_PROF_GATHER)  # the encoder gathers its pooled token rows with the same kernel: not K5
    db = finish_job(cv, emb, id0, n_total, sharded)
    del emb
    CTX.barrier()
    dt = CTX.all_max(time.perf_counter() - t0)
    ms, launches, nbytes = N.prof_read(N.SL_PROF_REDUCE)
    g_ms, g_n, g_bytes = (a - b for a, b in zip(N.prof_read(N.SL_PROF_GATHER), enc_g))  # K5 proper: embeds[sample_ids]
    N.prof_enable(False)
    if keep_db is not None:
        keep_db.update(db)
    del db
    gbps = nbytes / ms / 1e6 if ms else None
    # HBM bytes per launch from separate rocprofv3 --pmc passes over the same kernel instances (profiles/roofline_traffic.json
    # "legs": read = 2 x FETCH_SIZE x 1024 per MI355X_MICROARCH.md): the measured traffic / algorithmic ratio of the leg's kernel
    # applied to this run's algorithmic bytes per launch; not measured in this run, and said so
    traffic, traffic_source = None, None
    try:
        legs = json.loads((ROOT / "profiles" / "roofline_traffic.json").read_text()).get("legs", {})
        ratio = legs.get(traffic_key or "", {}).get("traffic_over_algorithmic")
        if ratio and launches:
            traffic = ratio * nbytes / launches
            traffic_source = (f"{legs.get('source')}: HBM read bytes / algorithmic bytes = {ratio} for this kernel at these shapes, "
                              "applied to this run's algorithmic bytes per launch")
    except Exception:
        pass
    out = {
        "workload": workload, "images_per_s": n_total / dt, "images": n_total, "n_gpus": world, "steps": steps, "batch": B,
        "reduce_cache_policy": policy_report(cv),
        "roofline": roofline_hbm(gbps, traffic=traffic, traffic_source=traffic_source, kernel=kernel, launches=launches,
                                 avg_launch_us=ms / max(launches, 1) * 1e3, algorithmic_bytes_per_launch=nbytes / max(launches, 1),
                                 condition="in-pipeline: inputs written by the model's last kernel microseconds earlier, "
                                           + ("embed on a second stream" if OVERLAP else "embed on the same stream")
                                           + (f"; rank 0 of {world} (every rank runs the same launches on its own shard)" if world > 1 else "")),
        # K5 (embeds[sample_ids], activation_based.py:387-390): C*k*D*4 bytes read + written per layer, one launch over the ids of all
        # layers.  The leg's (N, D) table is a few MB — far below the 256 MiB Infinity Cache — so the READ half is served by the cache
        # and only the WRITTEN half (C*k*D*4) is HBM traffic here: `achieved` prices the written bytes (at dataset scale,
        # 1.28 M x 512 = 2.6 GB, both halves are HBM traffic and the kernel's rate is `read_plus_written_GBps`)
        "gather_k5": {"bound": "hbm", "achieved": g_bytes / 2 / g_ms / 1e6 if g_ms else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                      "frac": g_bytes / 2 / g_ms / 1e6 / HBM_PEAK_GBPS if g_ms else None, "launches": g_n,
                      "written_bytes_per_launch": g_bytes / 2 / max(g_n, 1),
                      "read_plus_written_GBps": g_bytes / g_ms / 1e6 if g_ms else None,
                      "embedding_table_bytes": table_bytes,
                      "note": "HBM figure = written bytes / kernel time: the reads of the small per-leg embedding table hit the Infinity Cache"
                              + ("; sharded form (rows this rank owns, zeros elsewhere) + one all-reduce per layer" if sharded else "")},
    }
    if not args.no_self_check:
        out["self_check"] = self_check(dev, model, fm, args, n=check_n or 2 * B, B=B, layers=layers, agg=agg, cast=cast)
    return out


def _prompts(n):
    words = ["zebra", "stripe", "wheel", "sky", "grass", "dog", "cat", "red", "round", "metal", "wood", "water", "face", "text"]
    return [f"a photo of a {words[i % 14]} {words[(i // 14) % 14]} {i}" for i in range(n)]


@torch.no_grad()
def config3_leg(dev, args):
    """BASELINE configs[3] at its full geometry on one GPU: ViT-B/16 probed model (random init), ALL 12 encoder blocks
    (B,197,768) under aggregate_transformer_max (K2 + K3), SigLIP-so400m embed (27 x 1152, MLP 4304, patch 14, 256 tokens;
    `NativeSigLip` on the package's kernels) into concept_db (768, k, 1152) per block, then `Lens.text_probing` of 10,000
    prompts through the so400m TEXT tower (27 x 1152, ctx 64) against the 12 aggregated layers (K6: 92.16 M similarities)."""
    import numpy as np

    import oracle  # checker only
    from semanticlens_amd.foundation_models import NativeSigLip

    base = synth.SyntheticSigLip(device=dev)  # so400m geometry (synth.SIGLIP_SO400M)
    fm = NativeSigLip(base)
    vit = synth.vit_b16().to(dev)
    layers = [f"blocks.{i}" for i in range(12)]
    B = args.batch
    db = {}
    out = collect_leg(
        dev, fm, args, vit, layers, aggregators.aggregate_transformer_max,
        "colreduce2 (K2, (B, 197, 768) token activations, component axis contiguous; the 12 identical block outputs of a batch "
        "are reduced by ONE launch over a table of tensors once the last block has fired — 1.86 GB at B = 256 — and merged by "
        "one K3 launch; a collector's first batch runs layer by layer)",
        "BASELINE configs[3], full geometry: ViT-B/16 (random init) probed model, all 12 encoder blocks (B,197,768) fp32, "
        "aggregate_transformer_max, 7 262 208 B/image; embed = NativeSigLip at the SigLIP-so400m geometry (27 x 1152, 16 heads "
        "of 72, MLP 4304, patch 14 -> 256 tokens, D = 1152), on the same stream", steps=args.leg_steps or 8, B=B, check_n=B, overlap=False, keep_db=db,
        traffic_key="config3_full")
    assert all(v.shape == (768, args.k, 1152) for v in db.values()) and len(db) == 12
    agg_db = {name: v.mean(1) for name, v in db.items()}  # what a user hands text_probing (README: `concept_db[layer].mean(1)`)
    lens = Lens(fm, device=dev)
    prompts = _prompts(10000)
    if CTX.sharded:  # prompts sharded through the text tower, query rows through the cosine GEMM, result rows all-gathered
        def probe(p):
            return sld.text_probing_sharded(fm, p, agg_db, batch_size=1024)
    else:
        def probe(p):
            return lens.text_probing(p, agg_db, batch_size=1024)
    probe(prompts[: 1024 * CTX.world])  # warm-up
    wall, sims = CTX.timed(lambda: probe(prompts))
    assert all(v.shape == (10000, 768) for v in sims.values())
    n_sims = 10000 * 768 * 12
    # the probe against the oracle on a sample of the queries: the same device text embeddings, fp64 cosine on the host
    q = fm.encode_text(fm.tokenize(prompts[:64]).to(dev)).float().cpu().numpy()
    worst = 0.0
    for name in (layers[0], layers[-1]):
        want = oracle.similarity(q, agg_db[name].cpu().numpy())
        worst = max(worst, float(np.abs(sims[name][:64].cpu().numpy() - want).max()))
    if not worst < 1e-4:
        raise AssertionError(f"config3: text_probing differs from the oracle by {worst}")
    out["text_probing_from_prompts"] = {
        "queries_per_s": 10000 / wall, "Msim_per_s": n_sims / wall / 1e6, "wall_ms": wall * 1e3, "n_gpus": CTX.world,
        "workload": "Lens.text_probing: 10,000 prompts -> tokenizer -> so400m text tower (27 x 1152, ctx 64, batches of 1024) -> "
                    "cosine GEMM vs 12 layers x 768 components x D=1152 (the leg's own concept_db, mean over k)"
                    + ("; distributed.text_probing_sharded: prompts and query rows sharded over the ranks, embeddings and similarity "
                       "rows all-gathered, every rank holds the full result" if CTX.sharded else ""),
        "max_abs_diff_vs_oracle_64_queries": worst}
    out["embed_model"] = fm.name
    del db, agg_db, sims
    return out


def config4_leg(dev, fm, args):
    """`_config4_leg` under MIOpen's find mode (`torch.backends.cudnn.benchmark = True`, scoped to this leg).  ConvNeXt's block outputs
    are channels_last-strided (the residual add takes its permuted branch's layout; torchvision's and timm's blocks behave the
    same), and MIOpen's immediate-mode pick for an fp32 NHWC 7 x 7 depthwise convolution is its naive kernel — 60 % of the forward
    (profiles/r04_cfg4_leg_kernel_stats.csv).  Its search finds a kernel that makes the whole forward 1.8x faster (418 -> 229 ms at
    B = 256, tools/cfg4_miopen_find.py), which is what a user of this model on ROCm would switch on; the default-mode forward is
    timed beside it so that the leg's images/s can be read either way.  The headline keeps the driver's setting."""
    saved = torch.backends.cudnn.benchmark
    model = synth.convnext_l().to(dev)
    x = torch.randn(args.batch, 3, 224, 224, device=dev)

    def fwd_ms(reps=3):
        with torch.no_grad():
            model(x)
            model(x)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                model(x)
            torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    try:
        # PyTorch keeps ONE per-shape algorithm cache for both modes: a shape first run in immediate mode keeps its naive kernel
        # for the life of the process, so the search must be on before ConvNeXt's shapes are seen for the first time
        torch.backends.cudnn.benchmark = True
        ms_find = fwd_ms()
        del model, x
        out = _config4_leg(dev, fm, args)
    finally:
        torch.backends.cudnn.benchmark = saved
    out["miopen_find"] = {"enabled_for_this_leg": True, "convnext_l_forward_ms_find_mode": ms_find, "batch": args.batch,
                          "note": "torch.backends.cudnn.benchmark=True for this leg only: MIOpen's immediate-mode pick for the fp32 NHWC "
                                  "depthwise convolutions is its naive kernel (forward 416-418 ms at B = 256, this leg 597 images/s: "
                                  "tools/cfg4_miopen_find.py, profiles/r05_cfg4_miopen_find.txt)"}
    return out


def _config4_leg(dev, fm, args):
    """BASELINE configs[4] on one GPU: ConvNeXt-L (random init, 198 M parameters), the four stage outputs
    (192 x 56^2, 384 x 28^2, 768 x 14^2, 1536 x 7^2: 4 515 840 B/image) through (i) the activation collect (K1 + K3, roofline),
    (ii) the relevance-maximisation visualizer (EpsilonPlusFlat LRP backward in PyTorch, K1 sum + abs-norm + K3) and
    (iii) eval_clarity / eval_redundancy / eval_polysemanticity over the WHOLE concept_db (2 880 components x k x 512)."""
    import numpy as np

    import oracle  # checker only
    from semanticlens_amd import scores
    from semanticlens_amd.component_visualization import RelevanceComponentVisualizer

    model = synth.convnext_l().to(dev)
    layers = [f"stages.{i}" for i in range(4)]
    B = args.batch
    db = {}
    out = collect_leg(
        dev, fm, args, model, layers, aggregators.aggregate_conv_max,
        "colreduce2 (K1, channels_last fp32 stage outputs: (B, S, C) with S = 3136 / 784 / 196 / 49, C = 192 / 384 / 768 / 1536)",
        "BASELINE configs[4] collect stage: ConvNeXt-L (random init) probed model, stages.0-3 outputs fp32 NCHW, aggregate_conv_max, "
        "4 515 840 B/image (the stage outputs arrive channels_last — the residual add takes its permuted branch's layout — so K1 "
        "runs its component-contiguous kernel); embed = the headline's CLIP ViT-B/32, on the same stream", steps=args.leg_steps or 4, B=B, check_n=B,
        keep_db=db, overlap=False, traffic_key="config4_full")
    widths = (192, 384, 768, 1536)
    assert all(db[n].shape == (c, args.k, 512) for n, c in zip(layers, widths))
    # the same collect with block outputs written NCHW-contiguous (synth.ConvNeXtBlock nchw_out): the depthwise convolutions stay
    # on MIOpen's NCHW kernels instead of its naive NHWC one (60 % of the forward above) and K1 runs its row kernel
    model_nchw = synth.convnext_l(nchw_out=True).to(dev)
    model_nchw.load_state_dict(model.state_dict())
    nchw = collect_leg(
        dev, fm, args, model_nchw, layers, aggregators.aggregate_conv_max,
        "rowreduce (K1, NCHW-contiguous fp32 stage outputs: rows of S = 3136 / 784 / 196 / 49 floats)",
        "the same ConvNeXt-L weights with every block's residual sum written NCHW-contiguous (one transposing add per block)",
        steps=args.leg_steps or 4, B=B, check_n=B, overlap=False)
    out["nchw_block_outputs"] = {k_: nchw[k_] for k_ in ("workload", "images_per_s", "roofline", "reduce_cache_policy") if k_ in nchw}
    out["nchw_block_outputs"]["self_check"] = nchw.get("self_check")
    del model_nchw, nchw
    torch.cuda.empty_cache()

    # ---- (ii) relevance visualizer: forward + LRP backward per batch, both top-k states ----
    # N ranks: a (128 x N)-image set, rank r walks its contiguous shard, both sets of states are merged across the ranks
    # (distributed.run_sharded: one packed all-gather + K4 per set)
    world, sharded = CTX.world, CTX.sharded
    b_rel = 32
    n_rel = 128 * world
    u8 = synth.synth_images_u8(torch.arange(n_rel, device=dev))
    ds_model = _Rows(synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD).cpu(), f"cfg4-{n_rel}", True)
    ds_fm = _Rows(u8.cpu(), "cfg4-fm", False)
    first = {}

    def tapped(model_, modules, images, targets):
        from semanticlens_amd.component_visualization.lrp import lrp_epsilon_plus_flat

        res = lrp_epsilon_plus_flat(model_, modules, images, targets, epsilon=0.1, norm_pass=True)  # 1e-6 overflows on ConvNeXt-L (relevance_based.py)
        if not first:
            first.update({k_: (a.detach().clone(), r.detach().clone()) for k_, (a, r) in res.items()})
        return res

    with torch.enable_grad():
        cvr = RelevanceComponentVisualizer(model, ds_model, ds_fm, layers, num_samples=args.k, attribution=tapped, device=dev,
                                           **({"tie_mode": "total"} if sharded else {}))

        def rel_run():
            return sld.run_sharded(cvr, batch_size=b_rel) if sharded else cvr._run(batch_size=b_rel)

        rel_run()  # warm-up (MIOpen backward kernels)
        first.clear()
        dt_rel, _ = CTX.timed(rel_run)
    worst_rel = 0.0
    for name, c in zip(layers, widths):
        ids = cvr.get_max_reference(name)
        assert ids.shape == (c, args.k) and int(ids.max()) < n_rel and int(ids.min()) >= 0
        act, rel = first[name]
        got = cvr._summed(rel).cpu().numpy()
        want = oracle.agg_conv(rel.float().cpu().numpy(), "sum")
        if not (np.isfinite(want).all() and np.isfinite(got).all() and np.abs(want).max() > 0):
            raise AssertionError(f"config4: the relevance of {name} is not finite / all zero")
        scale = float(np.abs(want).max())
        worst_rel = max(worst_rel, float(np.abs(got - want).max()) / scale)
    if not worst_rel < 1e-5:
        raise AssertionError(f"config4: summed relevance differs from the oracle by {worst_rel} of its scale")
    out["relevance_visualizer"] = {
        "images_per_s": n_rel / dt_rel, "images": n_rel, "batch": b_rel, "n_gpus": world, "composite": cvr.composite,
        "workload": "RelevanceComponentVisualizer._run: ConvNeXt-L forward + EpsilonPlusFlat LRP backward (epsilon 0.1; PyTorch autograd), "
                    "relevance and activation of 4 stages -> K1 sum -> abs-norm -> K3 (two top-k states per layer)"
                    + ("; distributed.run_sharded: contiguous shards, both sets of states merged across the ranks" if sharded else ""),
        "summed_relevance_max_rel_diff_vs_oracle": worst_rel}
    del cvr, first

    # ---- (iii) the scores over the whole concept_db ----
    # N ranks: the concept_db is replicated (every rank holds the merged one); clarity and polysemanticity run with the component axis
    # sharded (distributed.eval_sharded: rank r scores rows shard_range(C_l, r, N) of every layer, one all-gather of the result rows);
    # redundancy is a C x C Gram matrix per layer: replicated
    lens = Lens(fm, device=dev)
    agg_db = {n: v.mean(1) for n, v in db.items()}

    def timed(fn, family):
        fn()  # warm-up
        N.prof_enable(True)
        N.prof_reset()
        wall, res = CTX.timed(fn)
        ms, launches, work = N.prof_read(family)
        N.prof_enable(False)
        return res, wall, ms, launches, work

    comps = sum(widths)
    in_bytes = comps * args.k * 512 * 4
    if sharded:
        cl, w_cl, ms_cl, n_cl, by_cl = timed(lambda: sld.eval_sharded(scores.clarity_score, db), N.SL_PROF_SCORES)
        po, w_po, ms_po, n_po, by_po = timed(lambda: sld.eval_sharded(scores.polysemanticity_score, db), N.SL_PROF_SCORES)
    else:
        cl, w_cl, ms_cl, n_cl, by_cl = timed(lambda: lens.eval_clarity(db), N.SL_PROF_SCORES)
        po, w_po, ms_po, n_po, by_po = timed(lambda: lens.eval_polysemanticity(db), N.SL_PROF_SCORES)
    rd, w_rd, ms_rd, n_rd, fl_rd = timed(lambda: lens.eval_redundancy(agg_db), N.SL_PROF_GEMM)
    # against the oracle: clarity of every component, polysemanticity of 48 components through scikit-learn, redundancy per layer
    worst = {"clarity": 0.0, "polysemanticity": 0.0, "redundancy": 0.0}
    equal_single = None
    if CTX.rank == 0:
        for name in layers:
            V = db[name].cpu().numpy()
            worst["clarity"] = max(worst["clarity"], float(np.abs(cl[name].cpu().numpy() - oracle.clarity(V)).max()))
            worst["redundancy"] = max(worst["redundancy"], abs(float(rd[name]) - float(oracle.redundancy(agg_db[name].cpu().numpy()))))
            sub = np.arange(0, V.shape[0], max(1, V.shape[0] // 12))[:12]
            worst["polysemanticity"] = max(worst["polysemanticity"],
                                           float(np.abs(po[name].cpu().numpy()[sub] - oracle.polysemanticity(V[sub])).max()))
        if not all(v < 1e-4 for v in worst.values()):
            raise AssertionError(f"config4: scores differ from the oracle: {worst}")
        if sharded:  # the sharded scores against the single-process call on this rank: every bit
            cl1, po1 = lens.eval_clarity(db), lens.eval_polysemanticity(db)
            equal_single = all(torch.equal(cl[n_], cl1[n_]) and torch.equal(po[n_], po1[n_]) for n_ in layers)
    out["scores_full_db"] = {
        "components": comps, "k": args.k, "D": 512, "input_bytes": in_bytes, "n_gpus": world,
        "clarity_k7": dict(roofline_hbm(by_cl / ms_cl / 1e6), kernel_ms=ms_cl, launches=n_cl, wall_ms=w_cl * 1e3,
                           components_per_s=comps / w_cl, bytes_this_rank=by_cl,
                           note="C*n*D*4 bytes read once; all four layers (2.4-19 MB each) in ONE sl_clarity_multi launch"
                                + (f"; component axis sharded over {world} ranks (kernel figures: rank 0's shard), wall = max over ranks "
                                   "incl. the all-gather of the result rows" if sharded else "")),
        "polysemanticity_k9": {"bound": "valu/lds", "components_per_s": comps / w_po, "kernel_ms": ms_po, "launches": n_po, "wall_ms": w_po * 1e3,
                               "input_GBps": by_po / ms_po / 1e6,
                               "note": "Gram matrix of each component from C*n*D*4 input bytes (read once), then sklearn's k-means++ / "
                                       "Lloyd / best-of-10 replayed in Gram space in fp64 in LDS: no HBM traffic per Lloyd iteration"
                                       + (f"; component axis sharded over {world} ranks, one all-gather of all layers' rows" if sharded else "")},
        "redundancy_k8": {"bound": "mfma", "achieved_TFLOPs": fl_rd / ms_rd / 1e9 if ms_rd else None, "kernel_ms": ms_rd, "launches": n_rd,
                          "wall_ms": w_rd * 1e3, "note": "K6 on (C,D)x(C,D) per layer + row max; C <= 1536: a few tiles, latency-bound"
                                                        + ("; replicated on every rank" if sharded else "")},
        "max_abs_diff_vs_oracle": worst,
        "sharded_equals_single_process": equal_single}
    del db, agg_db, model
    return out


def cpu_baseline_scores(threads_best: int):
    """Same-box CPU figures for the OTHER metric and the scores (BASELINE.md §2 holds container numbers): the reference's
    arithmetic on the host — `similarity_score` (scores.py:119-128: F.normalize + matmul = torch-CPU sgemm), `clarity_score`
    (scores.py:45-46), `polysemanticity_score` (scores.py:167-185: scikit-learn KMeans per component) — at SURVEY §6's shapes,
    with the thread count the images/s baseline picked AND with every core the process may use."""
    import numpy as np

    import oracle  # checker / baseline only

    try:
        all_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        all_cores = os.cpu_count()
    g = torch.Generator().manual_seed(6)
    x, y = torch.randn(10000, 1152, generator=g), torch.randn(768, 1152, generator=g)
    V = torch.randn(2048, 20, 512, generator=g)
    Vp = np.random.RandomState(4).randn(64, 20, 512).astype(np.float32)

    def best_of(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return min(ts)

    out = {"kind": "port", "threads_best": threads_best, "host_cores_affinity": all_cores,
           "sample": "similarity (10000,1152)x(768,1152)^T = one configs[3] layer (7.68 M similarities); clarity (2048,20,512); "
                     "polysemanticity 64 x (20,512) through scikit-learn; best of 3 after one warm-up call"}
    before = torch.get_num_threads()
    try:
        for tag, t in (("best_threads", threads_best), ("all_cores", all_cores)):
            torch.set_num_threads(t)
            oracle.set_threads(t)
            dt_sim = best_of(lambda: oracle.similarity_torch(x, y))
            dt_cl = best_of(lambda: oracle.clarity_torch(V))
            dt_cl_c = best_of(lambda: oracle.clarity(V.numpy()))
            out[tag] = {"threads": t, "similarity_Msim_per_s": 10000 * 768 / dt_sim / 1e6, "similarity_GFLOPs": 2 * 10000 * 768 * 1152 / dt_sim / 1e9,
                        "clarity_ms_2048x20x512": dt_cl * 1e3, "clarity_ms_oracle_c_openmp": dt_cl_c * 1e3}
        torch.set_num_threads(threads_best)
        oracle.set_threads(threads_best)  # scikit-learn's OpenMP loops thrash with one thread per logical CPU of a 256-CPU box
        oracle.polysemanticity(Vp[:4])
        t = time.perf_counter()
        oracle.polysemanticity(Vp)
        out["polysemanticity_ms_per_component"] = (time.perf_counter() - t) / 64 * 1e3
        out["polysemanticity_threads"] = threads_best
    finally:
        torch.set_num_threads(before)
    return out


@torch.no_grad()
def sharded_check(dev, model, fm, args, rank, world):
    """Multi-rank parity inside the measured run (never fatal: differences are REPORTED in the line).  Every rank collects one batch of
    `--batch` images (ids rank*B .. rank*B + B - 1) with oracle taps on its own forward pass (what `self_check` does at N = 1: the same
    device activations go through the product's hooks and through the oracle's aggregate + ActMax), then takes part in the cross-rank merge
    and the sharded gather.  The per-rank ORACLE states and embeddings travel to rank 0 as host objects; rank 0 merges them with the oracle's
    own `merge_states` / `gather_rows` and compares with the merged top-k bits, ids and concept_db the product holds — the merge and the
    gather across the wire against the CPU restatement, with no forward pass replayed (MIOpen is not run-to-run deterministic at every batch size)."""
    import numpy as np

    import oracle  # the checker — never on the product path

    B = args.batch
    n = world * B
    cv = make_cv(model, n, args.k, "total")
    refs, seen = {}, {name: rank * B for name in LAYERS}

    def tap(name):
        def fn(m, i, o):
            a = oracle.agg_conv(o.detach().float().cpu().numpy(), "max")
            if name not in refs:
                refs[name] = oracle.ActMaxOracle(args.k, a.shape[1], oracle.MODE_TOTAL)
            refs[name].update(a, np.arange(seen[name], seen[name] + a.shape[0]))
            seen[name] += a.shape[0]

        return fn

    modules = dict(model.named_modules())
    taps = [modules[name].register_forward_hook(tap(name)) for name in LAYERS]
    try:
        mine = [synth.synth_images_u8(torch.arange(rank * B, (rank + 1) * B, device=dev))]
        emb = run_steps(cv, fm, mine, rank * B, B)
    finally:
        for h in taps:
            h.remove()
    db = finish_job(cv, emb, rank * B, n, True)
    torch.cuda.synchronize()
    payload = ({name: (refs[name].vals.copy(), refs[name].ids.copy()) for name in LAYERS}, emb.float().cpu().numpy())
    gathered = [None] * world
    dist.all_gather_object(gathered, payload)
    if rank != 0:
        return None
    emb_all = np.concatenate([g[1] for g in gathered])
    out = {"images": n, "ranks": world, "topk_values_equal": True, "topk_ids_equal": True, "concept_db_equal": True}
    for name in LAYERS:
        ref = refs[name]
        if world > 1:
            ref.merge_states(np.stack([gathered[r][0][name][0] for r in range(1, world)]), np.stack([gathered[r][0][name][1] for r in range(1, world)]))
        am = cv.actmax_cache.cache[name]
        out["topk_values_equal"] &= bool(np.array_equal(am.activations.view(torch.int16).numpy().view(np.uint16), ref.vals))
        out["topk_ids_equal"] &= bool(np.array_equal(am.sample_ids.numpy(), ref.ids))
        out["concept_db_equal"] &= bool(np.array_equal(db[name].cpu().numpy(), oracle.gather_rows(emb_all, ref.ids)))
    out["note"] = ("per-rank oracle states (taps on each rank's own forward) merged by the oracle on rank 0 against the product's "
                   "all-gathered + K4-merged states and its sharded gather")
    return out


def _bench_sha16() -> str:
    import hashlib

    return hashlib.sha256(Path(__file__).read_bytes()).hexdigest()[:16]


def _hostname() -> str:
    import socket

    return socket.gethostname()


class _quiet_stdout:
    """File descriptor 1 points at stderr while this is active (and C stdio is flushed on both sides): RCCL prints a version banner
    with printf when a communicator comes up, and the contract is ONE JSON line on stdout."""

    def __enter__(self):
        import ctypes

        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves — one process per GPU, backend
    nccl (= RCCL) — by re-executing this file under torch.distributed.run on a free local port.  Rank 0 prints the ONE JSON
    line.  A box with fewer than N GPUs is refused loudly (SL_BENCH_SHARE_GPU=1: N ranks on GPU 0 over gloo — a debugging /
    test aid for the N > 1 code path, never a measurement)."""
    import socket

    have = torch.cuda.device_count()
    env = dict(os.environ)
    if have < args.gpus:
        if env.get("SL_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {have} HIP device(s) (one rank per GPU; "
                             "SL_BENCH_SHARE_GPU=1 runs the ranks on GPU 0 over gloo for testing)")
        env.setdefault("SL_BENCH_BACKEND", "gloo")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # SL_BENCH_BACKEND=gloo SL_BENCH_SHARE_GPU=1: debugging aid to exercise the N>1 code path with several
    # ranks on ONE GPU (collectives staged through the host); the driver's runs use nccl (RCCL), one GPU per rank.
    # SL_BENCH_FORCE_DIST=1: bring the process group up even for ONE rank, so that a 1-GPU box runs the N>1 code
    # (barriers, packed all-gather + K4, sharded K5 + all-reduce) over RCCL itself (tests/test_gpu_distributed.py).
    backend = os.environ.get("SL_BENCH_BACKEND", "nccl")
    if os.environ.get("SL_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or os.environ.get("SL_BENCH_FORCE_DIST") == "1"
    if sharded and "RANK" not in os.environ:  # SL_BENCH_FORCE_DIST without a launcher: a one-rank group on a free local port
        import socket

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if sharded:
        with _quiet_stdout():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.tie_mode is None:
        # one GPU: the reference's own tie order, so that "top-k indices bit-exact vs the reference" and the throughput figure
        # describe the same run; several GPUs: the total order (shard-invariant; the reference's order depends on batch AND shard cuts)
        args.tie_mode = "total" if (world > 1 or sharded) else "aten"
    if sharded and args.tie_mode != "total":
        raise SystemExit("a sharded build needs --tie-mode total (distributed.run_sharded)")
    comm = None  # the library's own RCCL communicator (None under gloo / SL_COLLECTIVES=torch)
    if sharded:
        with _quiet_stdout():
            comm = sld.native_comm(None, dev)
            if comm is not None:  # bring the communicator all the way up here (RCCL prints its banner on first use)
                comm.allreduce(torch.zeros(1, dtype=torch.float64, device=dev), "max")
                torch.cuda.synchronize()

    CTX.rank, CTX.world, CTX.sharded, CTX.backend, CTX.comm, CTX.dev = rank, world, sharded, backend, comm, dev
    all_max = CTX.all_max
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    global OVERLAP
    OVERLAP = bool(args.overlap)
    if args.quick:
        args.no_api_leg = args.no_channels_last = args.no_probing = args.no_cpu_baseline = True
        args.no_tokens_leg = args.no_half_leg = args.no_config4_leg = True

    B, K, W, bps = args.batch, args.steps, args.warmup, max(1, args.batches_per_step)
    if args.scaling == "strong":  # fixed TOTAL work, contiguous shards (distributed.shard_range), ids global
        n_total = args.images or K * bps * B
        id_start, id_stop = sld.shard_range(n_total, rank, world)
        n_local = id_stop - id_start
        n_batches = -(-n_local // B)
        K = -(-n_batches // bps)
    else:  # weak (the driver's contract): K steps of `bps` batches per rank
        n_batches = K * bps
        n_local = n_batches * B
        n_total = world * n_local
        id_start = rank * n_local
    model = synth.resnet50().to(dev)
    fm_base = synth.SyntheticClip(device=dev)
    fm = fm_base
    if args.fm != "torch":
        from semanticlens_amd.foundation_models.native_clip import NativeClip

        fm = NativeClip(fm_base, gemm="bf16x3" if args.fm == "native" else "f32")
    Lens(fm, device=dev)

    # ---- warm-up on throw-away state (MIOpen kernel selection, allocator, lazy kernel loads) ----
    # The warm-up runs the complete job (incl. the RCCL collectives of finish_job) on W steps = W x bps batches.
    warm_steps_run = 0
    if W:
        nw = W * bps
        warm = [synth.synth_images_u8(torch.arange(10**7 + i * B, 10**7 + (i + 1) * B, device=dev)) for i in range(nw)]
        # The W-step warm-up job is repeated until `--min-warmup-seconds` of device work have passed (every rank runs the
        # same count: the decision is all-reduced).  On a fresh box the first process measured 8 % below the second with
        # three warm-up batches only (5 423 vs 5 900 images/s): the part needs ~1 s of this load to settle, and MIOpen / the
        # code-object loader still have first-use work after three batches.  The timed region is untouched: exactly K steps.
        t_warm = None  # the clock starts after the first job: that one pays the one-time costs
        job_times = []
        while True:
            warm_cv = make_cv(model, world * nw * B, args.k, args.tie_mode)
            t_job = time.perf_counter()
            emb_w = run_steps(warm_cv, fm, warm, rank * nw * B, nw * B)
            finish_job(warm_cv, emb_w, rank * nw * B, world * nw * B, sharded)
            torch.cuda.synchronize()
            job_times.append(time.perf_counter() - t_job)
            warm_steps_run += W
            if t_warm is None:
                t_warm = time.perf_counter()
            spent = time.perf_counter() - t_warm
            # settled = the last two jobs ran within 3 % of the fastest one seen (a box that has just started can run the
            # first ten seconds of a process 10 % slow: 5 250 vs 5 850 images/s); give up waiting after ten times --min-warmup-seconds (15 s)
            settled = len(job_times) >= 3 and max(job_times[-2:]) <= 1.03 * min(job_times[1:])
            more = 1.0 if spent < args.min_warmup_seconds or (not settled and spent < 10.0 * args.min_warmup_seconds) else 0.0
            more = all_max(more)
            if not more:
                break
        del warm, emb_w, warm_cv

    # ---- inputs resident in HBM before the clock starts ------------------------------------------
    def batch_ids(s):
        return torch.arange(id_start + s * B, min(id_start + (s + 1) * B, id_start + n_local), device=dev)

    pool = args.pool_batches if 0 < args.pool_batches < n_batches else n_batches
    distinct = [synth.synth_images_u8(batch_ids(s)) for s in range(pool)]
    if pool < n_batches:  # cycle the resident pool; the last (possibly short) batch keeps its own size
        batches = [distinct[s % pool][: batch_ids(s).numel()] for s in range(n_batches)]
    else:
        batches = distinct

    assert n_local > 0, "every rank needs at least one sample (--images >= --gpus)"

    last_cv = [None]

    def timed_job(fm_used, batches, n_total, n_local, model_=None, id_start_=None, tie_mode=None, prof=True):
        """One complete job between barriers: collect + embed over `batches`, flush / cross-rank merge, concept_db gather."""
        cv_ = make_cv(model_ or model, n_total, args.k, tie_mode or args.tie_mode)
        last_cv[0] = cv_
        ids0 = id_start if id_start_ is None else id_start_
        N.prof_enable(prof)
        N.prof_reset()
        CTX.barrier()
        t0_ = time.perf_counter()
        embeds_ = run_steps(cv_, fm_used, batches, ids0, n_local)
        db_ = finish_job(cv_, embeds_, ids0, n_total, CTX.sharded)
        CTX.barrier()
        return time.perf_counter() - t0_, db_

    elapsed, concept_db = timed_job(fm, batches, n_total, n_local)
    headline_policy = policy_report(last_cv[0])
    red_ms, red_n, red_bytes = N.prof_read(N.SL_PROF_REDUCE)
    mrg_ms, mrg_n, _ = N.prof_read(N.SL_PROF_MERGE)
    gat_ms, gat_n, _ = N.prof_read(N.SL_PROF_GATHER)
    N.prof_enable(False)
    elapsed = all_max(elapsed)
    assert all(v.shape == (c, args.k, 512) for v, c in zip(concept_db.values(), (512, 1024, 2048)))
    del concept_db

    # ---- strong scaling: the north star's 1.28 M-image set in TOTAL, sharded over the ranks (every rank takes part) ----
    strong = None
    if args.strong_images > 0 and args.scaling == "weak" and not args.quick:
        s_total = args.strong_images
        s_start, s_stop = sld.shard_range(s_total, rank, world)
        s_local = s_stop - s_start
        assert s_local > 0, "--strong-images must be at least --gpus"
        s_nb = -(-s_local // B)
        s_pool = max(1, min(args.strong_pool_batches, s_nb, len(distinct)))
        # the resident pool is cycled (1.28 M images are 193 GB of uint8 pixels); ids stay unique and global
        s_batches = [distinct[i % s_pool][: min(B, s_local - i * B)] for i in range(s_nb)]
        # always in the shard-invariant `total` order — also at N = 1, where the headline runs `aten`: the N > 1 speed-ups are
        # taken against this figure, and a speed-up across two tie modes would compare two different K3 kernels
        s_elapsed, s_db = timed_job(fm, s_batches, s_total, s_local, id_start_=s_start, tie_mode="total", prof=False)
        s_elapsed = all_max(s_elapsed)
        assert all(v.shape == (c, args.k, 512) for v, c in zip(s_db.values(), (512, 1024, 2048)))
        del s_db, s_batches
        strong = {"images": s_total, "n_gpus": world, "seconds": s_elapsed, "images_per_s": s_total / s_elapsed,
                  "images_per_gpu": -(-s_total // world), "distinct_resident_batches": s_pool, "tie_mode": "total",
                  "workload": "the same job (collect + embed + merge + concept_db gather) over --strong-images samples in total, "
                              "contiguous shards of ceil(N / ranks) samples (distributed.shard_range), global sample ids"}

    shard_chk = None
    if sharded and not args.no_self_check:
        shard_chk = sharded_check(dev, model, fm, args, rank, world)

    # ---- the legs EVERY rank takes part in (N > 1: BASELINE configs[3] / [4] are 8-GPU configurations) -------------------------------
    # collect legs: per-rank shard + cross-rank merge + sharded gather; text_probing: prompts and query rows sharded;
    # scores: component axis sharded.  With one process and no process group the same legs run further down, in the order of
    # the earlier rounds' lines.
    multi = {}
    if sharded:
        if not args.no_tokens_leg:
            multi["config3_full"] = config3_leg(dev, args)
            torch.cuda.empty_cache()
        if not args.no_config4_leg:
            multi["config4_full"] = config4_leg(dev, fm, args)
            torch.cuda.empty_cache()
        if not args.no_probing:
            multi["text_probing"] = probing_leg(dev)
            multi["text_probing"]["from_prompts"] = probing_end_to_end(fm, dev)
        rccl_ranks = comm.info()[0] if comm is not None else (dist.get_world_size() if backend == "nccl" else None)
        pg_info = {"backend": backend, "world_size": dist.get_world_size()}
        if backend == "nccl" and comm is not None:
            assert comm.info()[0] == world == args.gpus, f"RCCL communicator spans {comm.info()[0]} ranks, --gpus {args.gpus}"
        # every rank leaves the process group together; what follows on rank 0 (single-process legs, the CPU baseline) needs no peer
        CTX.barrier()
        with _quiet_stdout():
            sld.destroy_native_comms()
            dist.destroy_process_group()
        CTX.sharded, CTX.comm = False, None
    if rank != 0:
        return

    traffic = None
    tpath = ROOT / "profiles" / "roofline_traffic.json"
    if tpath.exists():  # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_traffic.py)
        try:
            traffic = json.loads(tpath.read_text()).get("reduce_bytes_per_launch")
        except Exception:
            traffic = None
    achieved = red_bytes / red_ms / 1e6 if red_ms else None
    gemm_note = {"native": "split-bf16 x3 on the bf16 matrix cores, fp32 accumulate (fp32-class accuracy, 1e-6 on cosines)",
                 "native-f32": "fp32-input MFMA", "torch": "hipBLASLt fp32"}[args.fm]
    line = {
        "metric": "images/sec concept-db build",
        "value": n_total / elapsed,
        "unit": "images/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        # the arithmetic type of the path: every activation, reduction, embedding and score is fp32; the encoder's GEMM operands
        # are fp32 values carried as two bf16 halves (three MFMA products, fp32 accumulate: 8e-7 on cosines, tolerance 1e-4);
        # `images_per_s_fp32_gemm` is the same job with no bf16 operand anywhere
        "dtype": {"native": "f32 (encoder GEMMs split-bf16x3, fp32 accumulate)", "native-f32": "f32", "torch": "f32"}[args.fm],
        # the rule behind that parenthesis (tests/test_gpu_native_clip.py, profiles/r05_tower_accuracy.txt): a 3-product bf16 split
        # carries ~2^-16 relative error per product; un-normalised features land within ~1e-5 of their scale — 2-4e-5 absolute at
        # scale 3-4 with residual channels at 150-300 injected (bar: 1e-4 absolute vs float64), cosines 2-4e-7; `--fm native-f32`
        # (NativeClip(gemm="f32")) is the fp32-MFMA arithmetic at torch-fp32's own distance from float64 (4-7e-6)
        "dtype_note": "bf16x3 = fp32 operands as two bf16 halves, three MFMA products, fp32 accumulate: features within 1e-4 absolute of "
                      "float64 also with massive activations (2-4e-5 measured), cosines 2-4e-7; images_per_s_fp32_gemm is the same job in "
                      "fp32-MFMA arithmetic (no bf16 operand anywhere)",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: ResNet-50 (random init) layer2-4, synthetic 224x224 images, "
                        "CLIP ViT-B/32 (random init) embed, aggregate_conv_max",
            "images_total": n_total, "images_per_gpu": n_local, "batch": B, "num_samples_k": args.k,
            "batches_per_step": bps, "images_per_step": bps * B, "ms_per_batch": elapsed / n_batches * 1e3,
            # configs[1] names 50k images
            "full_config_size": n_total >= 50000,
            "warmup_steps_run": warm_steps_run,  # --warmup repeated for --min-warmup-seconds; all untimed
            "distinct_resident_batches": pool,
            "tie_mode": args.tie_mode,
            "streams": 2 if args.overlap else 1, "layers": LAYERS, "parallelism": f"shard{world}" if world > 1 else "single",
            "collectives": ("none" if not sharded else
                            "libsemanticlens_hip.so (RCCL behind the C ABI): sl_actmax_allgather_merge = pack + ONE ncclAllGather of all "
                            "layers' top-k states + K4, and one sl_comm_allreduce per layer of the sharded gather" if comm is not None else
                            f"torch.distributed ({backend}): one all_gather_into_tensor of the packed top-k states + one all_reduce per "
                            "layer of the sharded gather"),
            "rccl_world_size": rccl_ranks if sharded else None,
            "process_group": pg_info if sharded else None,
            "clip_encoder": {"native": "NativeClip (HIP kernels, split-bf16 x3 GEMMs, fp32-class accuracy)",
                             "native-f32": "NativeClip (HIP kernels, fp32-input MFMA GEMMs)",
                             "torch": "torch module (hipBLASLt fp32)"}[args.fm],
            "arithmetic": {"collect (K1/K3)": "fp32 max -> bf16 RNE candidates, exact", "probed model": "fp32 (PyTorch/MIOpen)",
                           "encoder GEMMs": gemm_note},
        },
        "roofline": roofline_hbm(
            achieved, traffic=traffic,
            # `traffic` is NOT measured in this run: PMC counters need their own rocprofv3 passes (tools/pmc_traffic.py)
            traffic_source="profiles/roofline_traffic.json (separate rocprofv3 --pmc passes of this command)" if traffic else None,
            kernel="rowreduce (K1, activation spatial-max -> bf16 candidates)",
            launches=red_n, avg_launch_us=red_ms / max(red_n, 1) * 1e3,
            algorithmic_bytes_per_launch=red_bytes / max(red_n, 1), reduce_cache_policy=headline_policy,
            condition="in-pipeline: inputs written by the model's last kernel microseconds earlier, encoder running on "
                      "a second stream" if args.overlap else "in-pipeline, single stream",
            # north_star asks >= 0.90 of the HBM roofline on the collect: NOT met against the 8 TB/s spec; this kernel runs at the
            # part's measured copy ceiling (MI355X_MICROARCH.md: 6.29 TB/s = 0.786 of spec), see `frac_of_measured_copy_ceiling`
            north_star_target={"frac": 0.90, "met_vs_spec": bool(achieved and achieved / HBM_PEAK_GBPS >= 0.90),
                               "met_vs_measured_copy_ceiling": bool(achieved and achieved / HBM_COPY_CEILING_GBPS >= 0.90)},
        ),
        "kernel_time_share": {
            "reduce_ms": red_ms, "merge_ms": mrg_ms, "merge_launches": mrg_n, "gather_ms": gat_ms,
            "gather_note": "the gather family also counts the encoder's pooled-token-row gathers (one small launch per encode); K5 proper "
                           "(embeds[sample_ids], one launch at the end of the job) is reported per leg as gather_k5",
            "collect_kernels_fraction_of_wall": (red_ms + mrg_ms) / (elapsed * 1e3),
            "collect_only_images_per_sec": n_local / ((red_ms + mrg_ms) / 1e3) if red_ms else None,
        },
    }
    # K3 (the streaming top-k merge) of the headline job: north_star's "activation top-k collect" is K1 + K3
    line["k3"] = {
        "kernel": ("actmax_update_aten_wave (one wavefront per component: libstdc++'s introselect / introsort steps evaluated by ballots, "
                   "ids bit-identical to torch.topk's CPU order); one launch per hooked layer and batch (SEMANTICLENS_AMD_BATCH_K3=1: ONE "
                   "launch per forward for all hooked layers — collect-only 1.46 -> 1.84 M images/s, end to end unchanged)"
                   if args.tie_mode == "aten" else
                   "actmax_merge (total order: value desc, id asc; one launch per --merge-every batches)"),
        "tie_mode": args.tie_mode, "launches": mrg_n, "avg_launch_us": mrg_ms / max(mrg_n, 1) * 1e3,
        "k1_avg_launch_us": red_ms / max(red_n, 1) * 1e3, "k1_launches": red_n,
        "k1_plus_k3_us_per_batch": (red_ms + mrg_ms) / max(n_batches, 1) * 1e3,
        "collect_only_images_per_s": n_local / ((red_ms + mrg_ms) / 1e3) if red_ms else None,
        "bytes_per_launch_note": "reads B*C*2 candidate bytes + 10*C*k state bytes: latency-bound, not a bandwidth kernel",
    }
    single = world == 1
    if shard_chk is not None:
        line["sharded_check"] = shard_chk
        # the analysis stage's own parity objects, next to the build's: the sharded probe and the sharded scores
        tp, sc = multi.get("text_probing", {}).get("sharded_check"), multi.get("config4_full", {}).get("scores_full_db")
        if tp is not None:
            shard_chk["probe_equal"] = bool(tp["equals_single_process_bitwise"] and tp["max_abs_diff_vs_oracle_64_queries"] < 1e-4)
        if sc is not None:
            shard_chk["scores_equal"] = bool(sc["sharded_equals_single_process"] and all(v < 1e-4 for v in sc["max_abs_diff_vs_oracle"].values()))
    if strong is not None:
        # the N = 1 figure the speed-up is taken against: measured by THIS command at --gpus 1 (its own `strong_scaling`
        # object; committed copy of the round's run: profiles/strong_scaling_n1.json)
        # N = 1 reference, in this order of preference: (i) this run (N = 1); (ii) the record an earlier `--gpus 1` run of THIS
        # command left on THIS box (/tmp, written below: the driver runs N = 1, 2, 4, 8 back to back); (iii) the committed copy of
        # the round's own N = 1 run (another box of the pool).  A reference in another tie order is never used.
        box_path = Path(os.environ.get("SL_BENCH_N1_RECORD", "/tmp/semanticlens_amd_strong_n1.json"))
        ref, ref_src = None, None
        for path, src in ((box_path, "an earlier `python bench.py --gpus 1` on this box (" + str(box_path) + ")"),
                          (ROOT / "profiles" / "strong_scaling_n1.json",
                           "profiles/strong_scaling_n1.json: a committed FILE (`strong_scaling` of `python bench.py --gpus 1` on another "
                           "MI355X of the pool), not measured in this run")):
            if ref is None and path.exists():
                try:
                    cand = json.loads(path.read_text())
                    ok = cand.get("images_per_s") and cand.get("tie_mode") == strong["tie_mode"] and cand.get("images") == strong["images"]
                    if ok and path == box_path:  # a record in /tmp counts only if THIS code wrote it on THIS box within the last 6 hours
                        ok = (cand.get("bench_sha16") == _bench_sha16() and cand.get("hostname") == _hostname()
                              and 0 <= time.time() - float(cand.get("written_at", 0)) < 6 * 3600)
                    if ok:
                        ref, ref_src = cand, src
                except Exception:
                    pass
        if world == 1 and not sharded:
            strong["speedup_vs_n1"] = 1.0
            strong["n1_reference"] = "this run"
            try:
                box_path.write_text(json.dumps(dict({k_: strong[k_] for k_ in ("images", "seconds", "images_per_s", "tie_mode", "n_gpus")},
                                                    bench_sha16=_bench_sha16(), hostname=_hostname(), written_at=time.time())))
            except OSError:
                pass
        elif ref is not None:
            strong["n1_reference"] = {"images_per_s": ref["images_per_s"], "images": ref.get("images"), "tie_mode": ref.get("tie_mode"),
                                      "source": ref_src}
            strong["speedup_vs_n1"] = strong["images_per_s"] / ref["images_per_s"]
        else:
            strong["n1_reference"] = None  # no N = 1 record in the same tie order: no speed-up is claimed
            strong["speedup_vs_n1"] = None
        strong["weak_scaling_images_per_s_same_run"] = n_total / elapsed
        line["strong_scaling"] = strong
    if single and not args.no_self_check:
        line["self_check"] = self_check(dev, model, fm, args)
    if single and not args.quick and not sharded:
        # both tie orders on the same 24 batches: 'aten' = torch.topk's CPU order on cat([state, batch]) per batch (one K3 launch per
        # layer and batch, ids bit-identical to the reference); 'total' = value desc / id asc, merged every 8 batches (the sharded mode)
        few = batches[: min(n_batches, 24)]
        n_few = sum(b.shape[0] for b in few)
        rates = {}
        for mode in ("total", "aten"):
            timed_job(fm, few[:2], 2 * B, 2 * B, tie_mode=mode, prof=False)
            dt_m, _ = timed_job(fm, few, n_few, n_few, tie_mode=mode, prof=False)
            rates[mode] = n_few / dt_m
        line["tie_modes"] = {"headline": args.tie_mode, "images_per_s": rates, "batches": len(few),
                             "weak_n1_total_order_images_per_s": rates["total"],  # what an N > 1 weak line (total order) compares with
                             "note": "aten: top-k ids bit-identical to the reference CPU path (activation_caching.py:133-141) at the "
                                     "same batch size; total: batch- and shard-invariant order used for N > 1"}
    if single and not args.quick:
        line["roofline"]["cold_inputs"] = reduce_cold_leg(dev, B)
    if single and not args.quick:
        # the two ceilings the headline sits under (north_star leaves the probed model's forward to PyTorch-ROCm): the same resident
        # batches through (i) the probed model alone — no hooks, no encoder — and (ii) the encoder alone
        few = [b for b in batches[: min(n_batches, 24)] if b.shape[0] == B]
        n_few = len(few) * B

        @torch.no_grad()
        def forward_only():
            for u8 in few:
                model(synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD))

        @torch.no_grad()
        def encode_only():
            emb_, filled_ = None, 0
            for u8 in few:
                emb_, filled_ = last_cv[0].embed_batch(fm, u8, emb_, filled_, n_few)

        for fn_ in (forward_only, encode_only):
            fn_()
        dt_f, _ = CTX.timed(forward_only)
        dt_e, _ = CTX.timed(encode_only)
        line["forward_only_images_per_s"] = n_few / dt_f
        line["encode_only_images_per_s"] = n_few / dt_e
        line["ceilings_note"] = (f"{len(few)} of the headline's batches: ResNet-50 forward alone (PyTorch/MIOpen fp32, hooks off) and the CLIP ViT-B/32 "
                                 "encode alone; run back to back on one stream they would give "
                                 f"{n_few / (dt_f + dt_e):.0f} images/s — the headline overlaps them on two streams and adds the collect")
        # SURVEY §8d sweep: batch size x num_samples, quick-length jobs of the headline's step (the reference's own defaults are
        # B = 32 / 64, activation_based.py:309,342; k = 20 in its tutorial, 100 in the relevance path)
        sweep = {}
        n_sw = min(n_local, 24 * B)
        flat = torch.cat(batches[: -(-n_sw // B)])[:n_sw]
        for b_ in (64, 256):
            parts = list(flat.split(b_))
            for k_ in (20, 100):
                saved_k = args.k
                args.k = k_
                try:
                    timed_job(fm, parts[: max(2, 512 // b_)], min(n_sw, 512), min(n_sw, 512), prof=False)  # first use of this batch size
                    dt_s, _ = timed_job(fm, parts, n_sw, n_sw, prof=False)
                finally:
                    args.k = saved_k
                sweep[f"B{b_}_k{k_}"] = n_sw / dt_s
        line["sweep"] = {"images_per_s": sweep, "images_per_job": n_sw, "tie_mode": args.tie_mode,
                         "note": "the headline's job (collect + embed on two streams + concept_db gather) at batch 64 / 256 and k = 20 / 100"}
    if single and args.fm == "native" and not args.quick:
        # the same job with every encoder GEMM on the fp32-input MFMA path (strict fp32 arithmetic end to end)
        from semanticlens_amd.foundation_models.native_clip import NativeClip

        few = batches[: min(n_batches, 24)]
        n_few = sum(b.shape[0] for b in few)
        fm32 = NativeClip(fm_base, gemm="f32")
        timed_job(fm32, few[:2], 2 * B, 2 * B, prof=False)  # first use of the fp32-MFMA kernels
        dt32, _ = timed_job(fm32, few, n_few, n_few)
        N.prof_enable(False)
        line["images_per_s_fp32_gemm"] = {"value": n_few / dt32, "batches": len(few),
                                          "note": "same step with NativeClip(gemm='f32'): no bf16 operand anywhere"}
    if single and not args.no_channels_last:
        # the same job on a channels_last copy of the probed model: the hooked activations arrive component-contiguous and
        # K1 runs its column-reduce kernel instead of the row-reduce kernel of the headline
        import copy

        model_cl = copy.deepcopy(model).to(memory_format=torch.channels_last)
        few = batches[: min(n_batches, 24)]
        n_few = sum(b.shape[0] for b in few)
        timed_job(fm, few[:2], 2 * B, 2 * B, model_cl)  # MIOpen picks its NHWC kernels
        dtcl, _ = timed_job(fm, few, n_few, n_few, model_cl)
        cl_ms, cl_n, cl_bytes = N.prof_read(N.SL_PROF_REDUCE)
        N.prof_enable(False)
        line["channels_last"] = {
            "images_per_s": n_few / dtcl, "batches": len(few),
            "k1": {"kernel": "colreduce (K1, component axis contiguous)", "GB/s": cl_bytes / cl_ms / 1e6 if cl_ms else None,
                   "frac": cl_bytes / cl_ms / 1e6 / HBM_PEAK_GBPS if cl_ms else None, "launches": cl_n,
                   "avg_launch_us": cl_ms / max(cl_n, 1) * 1e3},
            "self_check": None if args.no_self_check else self_check(dev, model_cl, fm, args),
            "note": "same step with model.to(memory_format=torch.channels_last); not the headline (the reference's models run NCHW)"}
        del model_cl
    if single and not args.no_half_leg:
        # fp16 copy of the probed model: the hooked activations are fp16 NCHW rows -> rowreduce_h (K1 for 2-byte elements)
        import copy

        model_h = copy.deepcopy(model).half()
        line["half_precision_model"] = collect_leg(
            dev, fm, args, model_h, LAYERS, aggregators.aggregate_conv_max,
            "rowreduce_h (K1 on fp16 NCHW activations: 16-byte pieces of 8 elements, fp32 compare, bf16 candidates)",
            "configs[1] with the probed ResNet-50 in fp16 (model.half(), fp16 inputs): layer2-4 activations are fp16, "
            "1 404 928 B/image", steps=min(n_batches, 16), B=B, cast=torch.float16, traffic_key="half_precision_model")
        del model_h
    line.update(multi)
    if single and not sharded and not args.no_tokens_leg:
        # BASELINE configs[3] at full geometry (ViT-B/16 x 12 blocks -> K2; so400m embed; 10k-prompt text_probing at D = 1152)
        line["config3_full"] = config3_leg(dev, args)
        torch.cuda.empty_cache()
    if single and not sharded and not args.no_config4_leg:
        # BASELINE configs[4]: ConvNeXt-L 4-stage collect (K1), relevance visualizer, scores over the whole concept_db
        line["config4_full"] = config4_leg(dev, fm, args)
        torch.cuda.empty_cache()
    if single and not args.no_api_leg:
        line["api_path"] = api_path_leg(dev, model, fm_base, args)
    if single and not sharded and not args.no_probing:
        line["text_probing"] = probing_leg(dev)
        line["text_probing"]["from_prompts"] = probing_end_to_end(fm, dev)
    if not args.no_cpu_baseline:  # rank 0 alone, any N: the other ranks have left (the process group is down), the GPUs are idle
        torch.manual_seed(0)
        line["cpu_baseline"] = cpu_baseline(args, synth.resnet50(), synth.SyntheticClip(device="cpu"))
        line["cpu_baseline"]["scores_and_probing"] = cpu_baseline_scores(line["cpu_baseline"]["threads"])
    print(json.dumps(line), flush=True)
    os.dup2(2, 1)  # whatever a library prints at exit goes to stderr


if __name__ == "__main__":
    main()
