"""A/B of library builds on the text_probing leg of bench.py (set SEMANTICLENS_AMD_LIB): GEMM time per call from sl_prof."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

out = bench.probing_leg(torch.device("cuda:0"))
r = out["roofline"]
print(json.dumps({"Gsim/s": round(out["value"] / 1e3, 1), "wall_ms": round(out["wall_ms"], 4), "gemm_avg_ms": round(r["avg_ms"], 4),
                  "gemm_frac": round(r["frac"], 4)}))
