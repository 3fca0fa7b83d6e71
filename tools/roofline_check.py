"""Recompute K1's roofline fraction from the committed rocprofv3 kernel-trace summaries alone, regime by regime.

  python tools/roofline_check.py profiles r03

fraction = algorithmic bytes per launch / AverageNs of the kernel's CSV row / 8 TB/s — no correction applied.  Beside it:
the figure the same command reports from its own HIP events WITHOUT the tool attached (what bench.py quotes), and the
difference of the two per-dispatch averages = what attaching the tool changes, measured.

Inputs under <dir>/ (tools/run_profiles_r03.sh):
  <R>_k1_cold_kernel_stats.csv           trace of tools/k1_cold_target.py         | <R>_k1_cold_events.json (no tool)
  <R>_reduce_pipe1_kernel_stats.csv      trace of tools/native/reduce_lab pipe1   | <R>_reduce_pipe1_lab.log (no tool)
  <R>_k1_inpipeline1s_kernel_stats.csv   trace of bench.py --quick --no-overlap   | <R>_bench_line_quick1s.json (no tool)
  <R>_k1_inpipeline_kernel_stats.csv     trace of bench.py --quick (two streams)  | <R>_bench_line_quick.json (no tool)
"""
import csv
import json
import os
import re
import sys

PEAK = 8.0e12
SHAPES = {"layer2": 256 * 512 * 784 * 4, "layer3": 256 * 1024 * 196 * 4, "layer4": 256 * 2048 * 49 * 4}
LAYERS = ("layer2", "layer3", "layer4")


def k1_rows(path):
    """K1 fp32 rows of a kernel-stats CSV -> {layer: (calls, average ns)}: layer2 is the G = 64 kernel, layer3 the aligned
    G = 16 one, layer4 the unaligned one (template arguments <float, G, U, OP, ALIGNED>)."""
    rows = {}
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            if "rowreduce_dma_kernel<float" not in n:
                continue
            args = n.split("rowreduce_dma_kernel<")[1].split(">")[0].replace(" ", "").split(",")
            g, aligned = int(args[1]), args[4] in ("true", "1")
            layer = "layer2" if g == 64 else ("layer3" if aligned else "layer4")
            calls, avg = int(r["Calls"]), float(r["AverageNs"])
            if layer in rows:
                c0, a0 = rows[layer]
                avg, calls = (a0 * c0 + avg * calls) / (c0 + calls), c0 + calls
            rows[layer] = (calls, avg)
    return rows


def section(title, rows, ref_us):
    """rows: {layer: (calls, avg ns)} from the CSV; ref_us: {layer: per-dispatch us without the tool} (or None)."""
    print(f"== {title}")
    tb = tt = tr = 0.0
    for layer in LAYERS:
        calls, avg = rows[layer]
        frac = SHAPES[layer] / (avg * 1e-9) / PEAK
        line = f"  {layer}: {calls:4d} calls, CSV average {avg / 1e3:6.2f} us -> {SHAPES[layer] / avg / 1e3:5.2f} TB/s = {frac:.3f} of 8 TB/s"
        if ref_us and layer in ref_us:
            line += f"   | no tool: {ref_us[layer]:6.2f} us = {SHAPES[layer] / (ref_us[layer] * 1e-6) / PEAK:.3f}   (tool - no tool: {avg / 1e3 - ref_us[layer]:+.2f} us)"
            tr += ref_us[layer] * 1e-6
        print(line)
        tb += SHAPES[layer]
        tt += avg * 1e-9
    out = tb / tt / PEAK
    print(f"  all layers (bytes / time): {out:.3f} from the CSV" + (f"   | {tb / tr / PEAK:.3f} without the tool   ratio {tr / tt:.3f}" if tr else ""))
    return out


def main(d, R):
    f = lambda name: os.path.join(d, f"{R}_{name}")  # noqa: E731
    if os.path.exists(f("k1_cold_kernel_stats.csv")):
        ev = json.load(open(f("k1_cold_events.json")))
        section("cold inputs (tools/k1_cold_target.py: rotated through > 1.2 GB, read-once policy)", k1_rows(f("k1_cold_kernel_stats.csv")),
                {la: ev[f"f32_{la}"]["avg_launch_us"] for la in LAYERS})
    if os.path.exists(f("reduce_pipe1_kernel_stats.csv")):
        ref = {}
        m = re.search(r"411 MB:\s+(\d+) GB/s.*?206 MB:\s+(\d+) GB/s.*?103 MB:\s+(\d+) GB/s", open(f("reduce_pipe1_lab.log")).read())
        if m:
            ref = {la: SHAPES[la] / (float(g) * 1e9) * 1e6 for la, g in zip(LAYERS, m.groups())}
        section("in-pipeline, native harness (tools/native/reduce_lab pipe1: in-place ReLU over the input, then K1, one stream, shipped policy)",
                k1_rows(f("reduce_pipe1_kernel_stats.csv")), ref)
    for tag, name in (("one stream (bench.py --quick --no-overlap)", "1s"), ("two streams, the headline's condition (bench.py --quick)", "")):
        csvp = f(f"k1_inpipeline{name}_kernel_stats.csv")
        if not os.path.exists(csvp):
            continue
        line = json.load(open(f(f"bench_line_quick{name}.json")))
        got = section(f"in-pipeline, the bench step: {tag}", k1_rows(csvp), None)
        print(f"  bench line without the tool: roofline.frac {line['roofline']['frac']:.3f} (avg launch {line['roofline']['avg_launch_us']:.2f} us)   "
              f"CSV / line = {got / line['roofline']['frac']:.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
