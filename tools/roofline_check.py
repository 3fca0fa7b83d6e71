"""Recompute `roofline.frac` of K1 from the committed rocprofv3 summaries alone.

  python tools/roofline_check.py profiles r03

Inputs (all under profiles/, produced by tools/run_profiles_r03.sh):
  <R>_k1_inpipeline_kernel_stats.csv   rocprofv3 --kernel-trace of `bench.py --quick` (K1 launches: in-pipeline regime only)
  <R>_k1_cold_kernel_stats.csv         rocprofv3 --kernel-trace of tools/k1_cold_target.py (cold regime only)
  <R>_bench_line_quick_profiled.json / _quick.json      the bench line of the same command with / without the tool attached
  <R>_k1_cold_events_profiled.json / _events.json       sl_prof HIP-event times of the cold launches with / without the tool
The tool's overhead per dispatch is MEASURED as (HIP-event average with the tool attached) - (without), per regime;
fraction = algorithmic bytes / (rocprof AverageNs - overhead) / 8 TB/s, compared with the un-profiled bench line.
"""
import csv
import json
import sys

PEAK = 8.0e12
SHAPES = {"layer2": 256 * 512 * 784 * 4, "layer3": 256 * 1024 * 196 * 4, "layer4": 256 * 2048 * 49 * 4}


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
            if layer in rows:  # several U variants of one layer: weight by calls
                c0, a0 = rows[layer]
                avg, calls = (a0 * c0 + avg * calls) / (c0 + calls), c0 + calls
            rows[layer] = (calls, avg)
    return rows


def main(d, R):
    quick = json.load(open(f"{d}/{R}_bench_line_quick.json"))
    quick_p = json.load(open(f"{d}/{R}_bench_line_quick_profiled.json"))
    ev = json.load(open(f"{d}/{R}_k1_cold_events.json"))
    ev_p = json.load(open(f"{d}/{R}_k1_cold_events_profiled.json"))
    print("== in-pipeline regime (bench.py --quick: K1 behind the model's last kernel, encoder on a second stream)")
    rows = k1_rows(f"{d}/{R}_k1_inpipeline_kernel_stats.csv")
    ov = quick_p["roofline"]["avg_launch_us"] - quick["roofline"]["avg_launch_us"]
    print(f"tool overhead per dispatch, measured: HIP-event average {quick_p['roofline']['avg_launch_us']:.2f} us with rocprofv3 attached "
          f"- {quick['roofline']['avg_launch_us']:.2f} us without = {ov:.2f} us")
    tb = tt = 0.0
    for layer in ("layer2", "layer3", "layer4"):
        calls, avg = rows[layer]
        t = avg * 1e-9 - ov * 1e-6
        print(f"  {layer}: {calls} calls, rocprof average {avg / 1e3:.2f} us -> corrected {t * 1e6:.2f} us -> {SHAPES[layer] / t / 1e12:.2f} TB/s "
              f"= {SHAPES[layer] / t / PEAK:.3f} of 8 TB/s")
        tb += SHAPES[layer]
        tt += t
    frac = tb / tt / PEAK
    print(f"  all layers: {frac:.3f}   bench line (no tool): {quick['roofline']['frac']:.3f}   ratio {frac / quick['roofline']['frac']:.3f}")
    print("== cold regime (tools/k1_cold_target.py: inputs rotated through > 1.2 GB, read-once policy)")
    rows = k1_rows(f"{d}/{R}_k1_cold_kernel_stats.csv")
    tb = tt = tb0 = tt0 = 0.0
    for layer in ("layer2", "layer3", "layer4"):
        ov = ev_p[f"f32_{layer}"]["avg_launch_us"] - ev[f"f32_{layer}"]["avg_launch_us"]
        calls, avg = rows[layer]
        t = avg * 1e-9 - ov * 1e-6
        print(f"  {layer}: {calls} calls, rocprof average {avg / 1e3:.2f} us, overhead {ov:.2f} us -> {SHAPES[layer] / t / 1e12:.2f} TB/s = "
              f"{SHAPES[layer] / t / PEAK:.3f}   HIP events, no tool: {ev[f'f32_{layer}']['frac_of_8TBps']:.3f}")
        tb += SHAPES[layer]
        tt += t
        tb0 += SHAPES[layer]
        tt0 += SHAPES[layer] / (ev[f"f32_{layer}"]["GB/s"] * 1e9)
    print(f"  all layers: {tb / tt / PEAK:.3f}   HIP events, no tool: {tb0 / tt0 / PEAK:.3f}   ratio {(tb / tt) / (tb0 / tt0):.3f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
