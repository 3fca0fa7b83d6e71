"""Only the native text tower (12 x 512, ctx 77, B = 1024, context-filling prompts), for rocprofv3 --kernel-trace --stats."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import synth  # noqa: E402
from semanticlens_amd.foundation_models.native_clip import NativeClip  # noqa: E402

base = synth.SyntheticClip(device="cuda:0")
fm = NativeClip(base)
long_prompt = " ".join(["zebra stripe wheel sky grass dog cat red round metal wood water face text"] * 8)
tok = base.tokenize([long_prompt] * 1024)
print("tokens per prompt:", int((tok[0] != 0).sum()))
for _ in range(8):
    fm.encode_text(tok)
torch.cuda.synchronize()
