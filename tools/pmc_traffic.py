"""profiles/rNN_pmc_{fetch,write}.csv (tools/prof_summarize.py pmc) -> profiles/roofline_traffic.json.

HBM bytes per launch of the hot kernels, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950:
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts exactly half of the bytes of a wide coalesced
streaming read (128-B requests tallied at 64 B), so the read side is doubled.  WRITE_SIZE is taken as is
(it matches the algorithmic B*C*2 bytes of the reduce kernels exactly).
    python tools/pmc_traffic.py profiles/r01_pmc_fetch.csv profiles/r01_pmc_write.csv profiles/roofline_traffic.json
"""
import csv
import json
import sys


def load(path, counter):
    out = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter and "sl::" in row["Kernel_Name"]:
                out[row["Kernel_Name"]] = float(row["MeanValue"])
    return out


def main(fetch_csv, write_csv, out_json):
    fetch = load(fetch_csv, "FETCH_SIZE")
    write = load(write_csv, "WRITE_SIZE")
    # bench shapes (tools/pmc_target.py): ResNet-50 layer2/3/4 at B=256 fp32, in kernel-template order
    # (the template list continues with the cache-policy argument, hence prefix matches)
    # round 2: the LDS-DMA kernels rowreduce_dma_kernel<G, U, OP, ALIGNED>: layer2 <64, 1, ..>, layer3 <16, 1, ..>, layer4 <16, 4, .., false>
    # round 3: the element type leads the template list, rowreduce_dma_kernel<float, G, U, OP, ALIGNED>
    # (later: further defaulted template arguments follow — MULTI, NI —, so the keys are prefixes without the closing bracket)
    algo = {"rowreduce_dma_kernel<float, 64, 1, 0, true,": 256 * 512 * 784 * 4, "rowreduce_dma_kernel<float, 16, 1, 0, true,": 256 * 1024 * 196 * 4,
            "rowreduce_dma_kernel<float, 16, 4, 0, false,": 256 * 2048 * 49 * 4,
            "rowreduce_dma_kernel<float, 64, 1, 0, true>": 256 * 512 * 784 * 4, "rowreduce_dma_kernel<float, 16, 1, 0, true>": 256 * 1024 * 196 * 4,
            "rowreduce_dma_kernel<float, 16, 4, 0, false>": 256 * 2048 * 49 * 4,
            "rowreduce_dma_kernel<64, 1, 0, true>": 256 * 512 * 784 * 4, "rowreduce_dma_kernel<16, 1, 0, true>": 256 * 1024 * 196 * 4,
            "rowreduce_dma_kernel<16, 4, 0, false>": 256 * 2048 * 49 * 4}
    kernels = {}
    tot_traffic = tot_algo = 0.0
    for name, kib in fetch.items():
        rd = 2.0 * kib * 1024.0
        wr = write.get(name, 0.0) * 1024.0
        entry = {"fetch_size_kib": kib, "write_size_kib": write.get(name), "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                 "hbm_bytes_per_launch": rd + wr}
        for key, a in algo.items():
            if key in name:
                entry["algorithmic_bytes_per_launch"] = a
                entry["traffic_over_algorithmic"] = (rd + wr) / a
                tot_traffic += rd + wr
                tot_algo += a
        kernels[name] = entry
    res = {
        "source": [fetch_csv, write_csv],
        "correction": "read = 2 x FETCH_SIZE x 1024 (gfx950 half-count of wide coalesced reads), write = WRITE_SIZE x 1024",
        "kernels": kernels,
        # mean over the three reduce launches of one bench step (layer2, layer3, layer4), like roofline.achieved
        "reduce_bytes_per_launch": tot_traffic / 3.0 if tot_algo else None,
        "reduce_algorithmic_bytes_per_launch": tot_algo / 3.0 if tot_algo else None,
    }
    with open(out_json, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: res[k] for k in ("reduce_bytes_per_launch", "reduce_algorithmic_bytes_per_launch")}))


if __name__ == "__main__":
    main(*sys.argv[1:4])
