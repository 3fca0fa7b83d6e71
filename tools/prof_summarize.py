"""Condense rocprofv3 CSV output into small summaries that fit in profiles/.

  python tools/prof_summarize.py trace  <dir> <out.csv>   # per-kernel count / total / avg / min / max (ns)
  python tools/prof_summarize.py pmc    <dir> <out.csv>   # per-kernel, per-counter mean value per dispatch
  python tools/prof_summarize.py context <dir> <out.txt> <substring>   # where in the kernel stream a kernel runs: its neighbours
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) <= 160 else name[:157] + "..."


def trace(d, out):
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    assert files, f"no kernel_trace.csv under {d}"
    agg = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values())
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v), round(100.0 * sum(v) / total, 3)])


def pmc(d, out):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    agg = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                agg[(short(row["Kernel_Name"]), row["Counter_Name"], row.get("Grid_Size", ""))].append(float(row["Counter_Value"]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "Dispatches", "MeanValue", "MinValue", "MaxValue"])
        for (k, c, g), v in sorted(agg.items()):
            w.writerow([k, g, c, len(v), sum(v) / len(v), min(v), max(v)])


def context(d, out, pattern):
    """Occurrences of kernels whose name contains `pattern`, in time order, each with the three kernels before and after it,
    its grid and its position in the run (fraction of the trace's span) — enough to name the leg / layer that launches it."""
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    assert files, f"no kernel_trace.csv under {d}"
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short(row["Kernel_Name"])[:110],
                             row.get("Grid_Size_X", row.get("Grid_Size", "")), row.get("Workgroup_Size_X", row.get("Workgroup_Size", ""))))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    hits = [i for i, r in enumerate(rows) if pattern in r[2]]
    with open(out, "w") as fh:
        fh.write(f"{len(hits)} launches of *{pattern}* among {len(rows)} kernels; total {sum(rows[i][1] - rows[i][0] for i in hits) / 1e6:.2f} ms "
                 f"of {sum(r[1] - r[0] for r in rows) / 1e6:.2f} ms of kernel time\n")
        # distinct (grid, workgroup) shapes and where in the run they sit
        shapes = defaultdict(list)
        for i in hits:
            shapes[(rows[i][3], rows[i][4])].append(i)
        for (g, wg), idx in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
            durs = [rows[i][1] - rows[i][0] for i in idx]
            pos = [(rows[i][0] - t0) / max(1, t1 - t0) for i in idx]
            fh.write(f"grid {g} wg {wg}: {len(idx)} launches, avg {sum(durs) / len(durs) / 1e3:.1f} us, run position {min(pos):.3f}..{max(pos):.3f}\n")
            i = idx[0]
            for j in range(max(0, i - 3), min(len(rows), i + 4)):
                fh.write(f"   {'>>' if j == i else '  '} {rows[j][2]}  [{(rows[j][1] - rows[j][0]) / 1e3:.1f} us]\n")


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc, "context": context}[sys.argv[1]](*sys.argv[2:])
