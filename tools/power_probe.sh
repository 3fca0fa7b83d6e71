#!/bin/bash
# Clock and power under sustained load (rocm-smi samples while a kernel loops): the split-bf16 cosine GEMM, the encoder's
# 160 x 256 GEMM (fc2 shape), and the K1 read stream, each for ~6 s.
cd "$GRAFT_REPO_ROOT"
cat > /tmp/loop.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from semanticlens_amd import _native as N
which = sys.argv[1]
if which == "cosine":
    q = torch.randn(10000, 1152, device="cuda:0"); y = torch.randn(9216, 1152, device="cuda:0")
    fn = lambda: N.similarity(q, y)
elif which == "fc2":
    a = N.Split.of(torch.randn(12800, 3072, device="cuda:0")); w = N.Split.of(torch.randn(768, 3072, device="cuda:0") * 0.02)
    out = torch.empty(12800, 768, device="cuda:0")
    fn = lambda: N.linear3(a, w, out=out)
elif which == "idle":
    fn = lambda: time.sleep(0.01)
else:
    xs = [torch.randn(256, 512, 28, 28, device="cuda:0") for _ in range(4)]
    cand = torch.empty(256, 512, dtype=torch.bfloat16, device="cuda:0")
    i = [0]
    def fn():
        i[0] += 1
        N.reduce_conv(xs[i[0] % 4], N.SL_CONV_MAX, cand, None)
t0 = time.time(); n = 0
while time.time() - t0 < 7.0:
    for _ in range(50): fn()
    torch.cuda.synchronize(); n += 50
print(which, "calls/s", n / (time.time() - t0))
PY
for w in idle cosine fc2 k1; do
  python /tmp/loop.py $w 2>/dev/null &
  pid=$!
  sleep 3.5
  for s in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Graphics\|Power (W)\|Average" | tr -s ' ' | head -4
    sleep 0.8
  done
  wait $pid
  echo "----"
done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
