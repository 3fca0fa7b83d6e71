"""rNN_pmc_half_{fetch,write}.csv (tools/prof_summarize.py pmc over tools/pmc_target_half.py) -> HBM bytes per launch of K1's
2-byte-element kernels against their algorithmic bytes, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE in KiB
counts half of the bytes of wide streaming reads: x 2 x 1024; WRITE_SIZE in KiB).  Launches are matched to shapes by size.
    python tools/pmc_traffic_half.py profiles/r06_pmc_half_fetch.csv profiles/r06_pmc_half_write.csv"""
import csv
import json
import sys

SHAPES = {"layer2 (256,512,28,28) fp16": 256 * 512 * 784 * 2, "layer3 (256,1024,14,14) fp16": 256 * 1024 * 196 * 2,
          "layer4 (256,2048,7,7) fp16": 256 * 2048 * 49 * 2}


def load(path, counter):
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter and "reduce" in r["Kernel_Name"] and "sl" in r["Kernel_Name"]:  # (_Float16 instances stay mangled)
                rows.append((r["Kernel_Name"], r.get("Grid_Size", ""), float(r["MeanValue"])))
    return rows


def main(fetch_csv, write_csv):
    fetch = load(fetch_csv, "FETCH_SIZE")
    write = {(k, g): v for k, g, v in load(write_csv, "WRITE_SIZE")}
    out, tot_rd, tot_algo = {}, 0.0, 0.0
    for name, grid, kib in fetch:
        rd = 2.0 * kib * 1024.0
        shape, algo = min(SHAPES.items(), key=lambda kv: abs(kv[1] - rd))
        wr = write.get((name, grid), 0.0) * 1024.0
        out[shape] = {"kernel": name.split("(")[0][:90], "hbm_read_bytes": rd, "hbm_write_bytes": wr, "algorithmic_bytes": algo,
                      "ratio": (rd + wr) / algo}
        tot_rd += rd + wr
        tot_algo += algo
    out["traffic_over_algorithmic"] = round(tot_rd / tot_algo, 4) if tot_algo else None
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
