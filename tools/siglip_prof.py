"""Only the native SigLIP-so400m image tower (27 x 1152, 256 tokens; random init) at B = argv[1] (default 64), 10 encodes, for
`rocprofv3 --kernel-trace --stats`: per-kernel times of configs[3]'s embed model."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth  # noqa: E402
from semanticlens_amd.foundation_models import NativeSigLip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
base = synth.SyntheticSigLip(device="cuda:0", t_layers=1)
fm = NativeSigLip(base)
img = torch.randn(B, 3, 224, 224, device="cuda:0")
for _ in range(10):
    fm.encode_image(img)
torch.cuda.synchronize()
