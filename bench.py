"""bench.py — concept-DB build throughput (BASELINE.json configs[1]) + text_probing, on N MI355X.

    python bench.py --gpus 1 --steps 196 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of B synthetic images already resident in HBM:
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=196)  # 196 x 256 = 50,176 images (configs[1]: "50k")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=20)  # num_samples; the reference tutorial's value
    ap.add_argument("--tie-mode", default="total", choices=["total", "aten"])
    ap.add_argument("--cpu-images", type=int, default=192, help="bounded sample for the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probing", action="store_true")
    ap.add_argument("--fm", default="native", choices=["native", "native-f32", "torch"],
                    help="CLIP ViT-B/32 encoder: package kernels (split-bf16 x3 or fp32 MFMA GEMMs) or the torch module")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="run the CLIP embed of each batch on the same HIP stream as forward + collect (default: a second "
                         "stream, +6 %% images/s; K1 then shares HBM with the encoder: in-bench roofline fraction 0.737 vs 0.75)")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (exhaustive MIOpen search)")
    return ap.parse_args()


class _Len:
    """Length-only dataset stand-in: the timed loop feeds device-resident batches directly."""

    def __init__(self, n, name):
        self.n, self.name = n, name

    def __len__(self):
        return self.n


def make_cv(model, n_total, k, tie_mode):
    return ActivationComponentVisualizer(
        model, _Len(n_total, f"synthetic-{n_total}"), _Len(n_total, "synthetic-fm"), LAYERS, num_samples=k,
        aggregate_fn=aggregators.aggregate_conv_max, cache_dir=None, tie_mode=tie_mode,
    )


OVERLAP = True  # embed on a second HIP stream beside forward + collect (--no-overlap: one stream)


@torch.no_grad()
def run_steps(cv, fm, batches, id_start, n_local):
    """The timed inner loop: K steps over device-resident uint8 batches."""
    for name in LAYERS:
        cv.actmax_cache.sample_idx_counter[name] = id_start
    embeds, filled = None, 0
    overlap = OVERLAP
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream() if overlap else None
    with cv.actmax_cache.hook_context(cv.model):
        for u8 in batches:
            if overlap:  # hot loop 2 (embed) on a second HIP stream beside hot loop 1 (forward + collect)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    embeds, filled = cv.embed_batch(fm, u8, embeds, filled, n_local)
                cv.collect_batch(synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD))
            else:
                cv.collect_batch(synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD))
                embeds, filled = cv.embed_batch(fm, u8, embeds, filled, n_local)
    if overlap:
        main.wait_stream(side)
    return embeds


@torch.no_grad()
def finish_job(cv, embeds, id_start, n_total, world):
    """Flush/merge the top-k states and gather the concept_db (device tensors)."""
    if world > 1:
        sld.merge_actmax_cache(cv.actmax_cache)
        return {n: sld.gather_concept_db_sharded(embeds, id_start, n_total, cv.get_max_reference(n)) for n in LAYERS}
    return {n: N.gather_rows(embeds, cv.actmax_cache.cache[n].device_state()[1]) for n in LAYERS}


def cpu_baseline(args, model_cpu, fm_cpu):
    """The CPU path (oracle = port of the reference's arithmetic) on a bounded sample of the same
    workload, on this box's host cores: torch-CPU forward under hooks -> oracle aggregate ->
    oracle ActMax.update (torch.topk tie order) ; CLIP encode on CPU ; oracle gather."""
    import numpy as np

    import oracle  # checker / baseline only — never on the product path

    B = 64
    # pick the torch thread count that is fastest on this box (containers often expose more logical CPUs
    # than their quota sustains: on the MI355X boxes 32 threads beat 128 by 3x)
    probe = synth.normalize_u8(synth.synth_images_u8(torch.arange(16)), synth.IMAGENET_MEAN, synth.IMAGENET_STD)
    best_t, best_dt = torch.get_num_threads(), float("inf")
    cands = sorted({t for t in (8, 16, 32, 64, torch.get_num_threads()) if t <= torch.get_num_threads()})
    with torch.no_grad():
        for t in cands:
            torch.set_num_threads(t)
            model_cpu(probe)
            t0 = time.perf_counter()
            model_cpu(probe)
            dt = time.perf_counter() - t0
            if dt < best_dt:
                best_t, best_dt = t, dt
    torch.set_num_threads(best_t)
    threads = best_t
    oracle.set_threads(threads)
    n = max(B, (args.cpu_images // B) * B)
    states = {}
    grabbed = {}
    agg_s = [0.0]

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
    for n_ in LAYERS:
        grabbed[n_] = 0
    embeds = []
    with torch.no_grad():  # one untimed batch: thread pools, oneDNN primitive caches
        u8 = synth.synth_images_u8(torch.arange(10**7, 10**7 + B))
        model_cpu(synth.normalize_u8(u8, synth.IMAGENET_MEAN, synth.IMAGENET_STD))
        fm_cpu.encode_image(fm_cpu.preprocess(u8))
    states.clear()
    for n_ in LAYERS:
        grabbed[n_] = 0
    agg_s[0] = 0.0
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
    for h in handles:
        h.remove()
    bytes_per_img = 2809856  # ResNet-50 layer2+3+4 fp32 activations per image
    return {
        "value": n / dt, "unit": "images/s", "cores": threads, "kind": "port",
        "sample": f"{n} synthetic images (batch {B}), same models/layers/k, torch-CPU forward + oracle collect "
                  f"(ATen tie order) + CPU CLIP encode + gather; {dt:.1f} s",
        "collect_only_GBps": n * bytes_per_img / agg_s[0] / 1e9,
        "collect_only_seconds": agg_s[0],
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
    lens.text_probing(prompts[:1024], db, batch_size=1024)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = lens.text_probing(prompts, db, batch_size=1024)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert all(v.shape == (10000, 768) for v in out.values())
    return {"queries_per_s": 10000 / wall, "wall_ms": wall * 1e3,
            "workload": "10,000 prompts -> tokenizer -> text tower (12 x 512, ctx 77, batches of 1024) -> probe vs 12 x 768 x 512"}


def probing_leg(dev):
    """text_probing at BASELINE configs[3] shapes: Q=10,000 query embeddings (D=1152, SigLIP-so400m width) against
    12 layers x 768 components, through `_probe` (lens.py:206-214).  The cosine GEMM (K6) is timed per dispatch
    through sl_prof.  Measured in both arithmetic modes; the default (split-bf16 x3) is the headline."""
    g = torch.Generator(device=dev).manual_seed(2)
    Q, D, C, L = 10000, 1152, 768, 12
    q = torch.randn(Q, D, device=dev, generator=g)
    db = {f"block{i}": torch.randn(C, D, device=dev, generator=g) for i in range(L)}
    from semanticlens_amd.lens import _probe

    def run(mode):
        N.set_gemm_mode(mode)
        _probe(q, db)  # warm-up
        torch.cuda.synchronize()
        N.prof_enable(True)
        N.prof_reset()
        t0 = time.perf_counter()
        out = _probe(q, db)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms, launches, flops = N.prof_read(N.SL_PROF_GEMM)
        N.prof_enable(False)
        assert all(v.shape == (Q, C) for v in out.values())
        return wall, ms, launches, flops, out

    wall3, ms3, n3, fl3, out3 = run("bf16x3")
    wall1, ms1, n1, fl1, out1 = run("f32")
    N.set_gemm_mode(None)
    max_diff = max((out3[k] - out1[k]).abs().max().item() for k in out3)
    sims = Q * C * L
    return {
        "metric": "Msimilarities/sec text_probing", "value": sims / wall3 / 1e6, "unit": "Msim/s",
        "workload": f"Q={Q} x {L} layers x C={C}, D={D} (configs[3] shapes), query embeddings resident",
        "wall_ms": wall3 * 1e3,
        # split-bf16 x3: three bf16 MFMAs per product.  `achieved` counts ALGORITHMIC flops (2*Q*C*D); the peak it is
        # priced against is the dense bf16 MFMA peak divided by the 3 products (2500 / 3); 3x achieved is what the
        # matrix cores actually issue.
        "roofline": {"bound": "mfma", "achieved": fl3 / ms3 / 1e9, "peak": MFMA_BF16_PEAK_TFLOPS / 3, "unit": "TFLOP/s",
                     "frac": (fl3 / ms3 / 1e9) / (MFMA_BF16_PEAK_TFLOPS / 3),
                     "kernel": "gemm3_nt_dma256 (split-bf16 x3 on v_mfma_f32_32x32x16_bf16, 256x128 tiles staged by LDS-DMA, all 12 layers in one launch; fp32-class accuracy)",
                     "mfma_flops_issued_TFLOPs": 3 * fl3 / ms3 / 1e9,
                     "ratio_to_fp32_mfma_peak": (fl3 / ms3 / 1e9) / MFMA_F32_PEAK_TFLOPS,
                     "launches": n3, "avg_ms": ms3 / max(n3, 1)},
        # the same probe with the exact-fp32 kernel (SL_GEMM_MODE=f32)
        "fp32_mfma_mode": {"value": sims / wall1 / 1e6, "unit": "Msim/s", "wall_ms": wall1 * 1e3,
                           "roofline": {"bound": "mfma", "achieved": fl1 / ms1 / 1e9, "peak": MFMA_F32_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": (fl1 / ms1 / 1e9) / MFMA_F32_PEAK_TFLOPS,
                                        "kernel": "gemm_nt (v_mfma_f32_32x32x2_f32)", "launches": n1,
                                        "avg_ms": ms1 / max(n1, 1)}},
        "max_abs_diff_between_modes": max_diff,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # SL_BENCH_BACKEND=gloo SL_BENCH_SHARE_GPU=1: debugging aid to exercise the N>1 code path with several
    # ranks on ONE GPU (collectives staged through the host); the driver's runs use nccl (RCCL), one GPU per rank.
    backend = os.environ.get("SL_BENCH_BACKEND", "nccl")
    if os.environ.get("SL_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    global OVERLAP
    OVERLAP = bool(args.overlap)

    B, K, W = args.batch, args.steps, args.warmup
    n_local = K * B
    n_total = world * n_local
    id_start = rank * n_local
    model = synth.resnet50().to(dev)
    fm = synth.SyntheticClip(device=dev)
    if args.fm != "torch":
        from semanticlens_amd.foundation_models.native_clip import NativeClip

        fm = NativeClip(fm, gemm="bf16x3" if args.fm == "native" else "f32")
    Lens(fm, device=dev)

    # ---- warm-up on throw-away state (MIOpen kernel selection, allocator, lazy kernel loads) ----
    # The warm-up runs the complete job (incl. the RCCL collectives of finish_job) on W batches.
    if W:
        warm_cv = make_cv(model, world * W * B, args.k, args.tie_mode)
        warm = [synth.synth_images_u8(torch.arange(10**7 + i * B, 10**7 + (i + 1) * B, device=dev)) for i in range(W)]
        emb_w = run_steps(warm_cv, fm, warm, rank * W * B, W * B)
        finish_job(warm_cv, emb_w, rank * W * B, world * W * B, world)
        del warm, emb_w, warm_cv

    # ---- inputs resident in HBM before the clock starts ------------------------------------------
    batches = [synth.synth_images_u8(torch.arange(id_start + s * B, id_start + (s + 1) * B, device=dev)) for s in range(K)]
    cv = make_cv(model, n_total, args.k, args.tie_mode)
    N.prof_enable(True)
    N.prof_reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    embeds = run_steps(cv, fm, batches, id_start, n_local)
    concept_db = finish_job(cv, embeds, id_start, n_total, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    red_ms, red_n, red_bytes = N.prof_read(N.SL_PROF_REDUCE)
    mrg_ms, mrg_n, _ = N.prof_read(N.SL_PROF_MERGE)
    gat_ms, gat_n, _ = N.prof_read(N.SL_PROF_GATHER)
    N.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(v.shape == (c, args.k, 512) for v, c in zip(concept_db.values(), (512, 1024, 2048)))

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    traffic = None
    tpath = ROOT / "profiles" / "roofline_traffic.json"
    if tpath.exists():  # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_traffic.py)
        try:
            traffic = json.loads(tpath.read_text()).get("reduce_bytes_per_launch")
        except Exception:
            traffic = None
    achieved = red_bytes / red_ms / 1e6 if red_ms else None
    line = {
        "metric": "images/sec concept-db build",
        "value": n_total / elapsed,
        "unit": "images/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: ResNet-50 (random init) layer2-4, synthetic 224x224 images, "
                        "CLIP ViT-B/32 (random init) embed, aggregate_conv_max",
            "images_per_gpu": n_local, "batch": B, "num_samples_k": args.k, "tie_mode": args.tie_mode,
            "streams": 2 if args.overlap else 1, "layers": LAYERS, "parallelism": f"shard{world}" if world > 1 else "single",
            "clip_encoder": {"native": "NativeClip (HIP kernels, split-bf16 x3 GEMMs, fp32-class accuracy)",
                             "native-f32": "NativeClip (HIP kernels, fp32-input MFMA GEMMs)",
                             "torch": "torch module (hipBLASLt fp32)"}[args.fm],
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": traffic,
            "kernel": "rowreduce (K1, activation spatial-max -> bf16 candidates)",
            "launches": red_n, "avg_launch_us": red_ms / max(red_n, 1) * 1e3,
            "algorithmic_bytes_per_launch": red_bytes / max(red_n, 1),
        },
        "kernel_time_share": {
            "reduce_ms": red_ms, "merge_ms": mrg_ms, "merge_launches": mrg_n, "gather_ms": gat_ms,
            "collect_kernels_fraction_of_wall": (red_ms + mrg_ms) / (elapsed * 1e3),
            "collect_only_images_per_sec": n_local / ((red_ms + mrg_ms) / 1e3) if red_ms else None,
        },
    }
    if world == 1 and not args.no_probing:
        line["text_probing"] = probing_leg(dev)
        line["text_probing"]["from_prompts"] = probing_end_to_end(fm, dev)
    if world == 1 and not args.no_cpu_baseline:
        torch.manual_seed(0)
        line["cpu_baseline"] = cpu_baseline(args, synth.resnet50(), synth.SyntheticClip(device="cpu"))
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
