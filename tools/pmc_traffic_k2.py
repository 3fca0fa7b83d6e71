"""profiles/r04_pmc_k2_{fetch,write}.csv (tools/prof_summarize.py pmc over tools/pmc_target_k2.py) -> HBM bytes per launch of
colreduce2 against its algorithmic bytes, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE in KiB counts half
of the bytes of wide streaming reads: x 2 x 1024; WRITE_SIZE in KiB).  Rows are matched to shapes by grid size (tasks x 64 x waves).
    python tools/pmc_traffic_k2.py profiles/r04_pmc_k2_fetch.csv profiles/r04_pmc_k2_write.csv
"""
import csv
import sys

SHAPES = [((12 * 256, 197, 768), 4), ((256, 197, 768), 4), ((256, 197, 768), 2), ((256, 3136, 192), 4), ((256, 784, 384), 4), ((256, 196, 768), 4), ((256, 49, 1536), 4)]


def load(path, counter):
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter and "colreduce2_kernel" in r["Kernel_Name"]:
                rows.append((r["Kernel_Name"], r["Grid_Size"], int(r["Dispatches"]), float(r["MeanValue"])))
    return rows


def main(fetch_csv, write_csv):
    fetch, write = load(fetch_csv, "FETCH_SIZE"), {(k, g): v for k, g, _, v in load(write_csv, "WRITE_SIZE")}
    algos = sorted({(s[0] * s[1] * s[2] * e, s, e) for s, e in SHAPES})
    print("kernel instance | grid | dispatches | HBM read bytes / launch (2 x FETCH_SIZE x 1024) | nearest algorithmic input | ratio | write bytes")
    for name, grid, n, kib in sorted(fetch, key=lambda r: -r[3]):
        rd = 2.0 * kib * 1024.0
        algo, shape, e = min(algos, key=lambda a: abs(a[0] - rd))
        short = name.split("colreduce2_kernel")[1].split("(")[0]
        print(f"colreduce2{short} | {grid} | {n} | {rd / 1e6:9.2f} MB | {shape} x {e} B = {algo / 1e6:9.2f} MB | {rd / algo:5.3f} | "
              f"{write.get((name, grid), 0.0) * 1024 / 1e6:6.3f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:3])
