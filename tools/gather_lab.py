"""K5 (`sl_gather_rows`) at the bench's shapes: kernel time from the dispatch's own timestamps (sl_prof), bit-equality with
torch's advanced indexing, and the rate on WRITTEN bytes (what `gather_k5.frac` prices) and on read + written bytes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semanticlens_amd import _native as N  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
for tag, n_rows, D, n_ids in (("configs[3] leg: (2 048, 1152) table, 12 x 768 x 20 ids", 2048, 1152, 9216 * 20),
                              ("configs[1]: (51 200, 512) table, 3 584 x 20 ids", 51200, 512, 3584 * 20),
                              ("configs[4] leg: (1 024, 512) table, 2 880 x 20 ids", 1024, 512, 2880 * 20),
                              ("configs[2]: (1 281 167, 512) table = 2.6 GB, 3 584 x 20 ids", 1281167, 512, 3584 * 20),
                              ("D = 768, k = 100: (50 000, 768) table, 3 584 x 100 ids", 50000, 768, 3584 * 100),
                              ("odd: (1 000, 20) table, 5 000 ids", 1000, 20, 5000)):
    emb = torch.randn(n_rows, D, device=DEV, generator=g)
    ids = torch.randint(-1, n_rows, (n_ids,), device=DEV, generator=g)
    out = N.gather_rows(emb, ids)
    want = emb[ids]
    ok = torch.equal(out, want)
    del want
    N.prof_enable(True)
    N.prof_reset()
    for _ in range(10):
        N.gather_rows(emb, ids)
    torch.cuda.synchronize()
    ms, launches, work = N.prof_read(N.SL_PROF_GATHER)
    N.prof_enable(False)
    us = ms / launches * 1e3
    wr = n_ids * D * 4
    print(f"{tag}: {us:8.1f} us | written {wr / 1e6:7.1f} MB -> {wr / us / 1e3:6.0f} GB/s ({wr / us / 1e3 / 8000:.3f} of 8 TB/s) | read + written "
          f"{2 * wr / us / 1e3:6.0f} GB/s | equal to emb[ids]: {ok}", flush=True)
    del emb, ids, out
