"""Condense rocprofv3 CSV output into small summaries that fit in profiles/.

  python tools/prof_summarize.py trace  <dir> <out.csv>   # per-kernel count / total / avg / min / max (ns)
  python tools/prof_summarize.py pmc    <dir> <out.csv>   # per-kernel, per-counter mean value per dispatch
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


if __name__ == "__main__":
    {"trace": trace, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
