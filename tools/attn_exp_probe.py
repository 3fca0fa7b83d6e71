import sys, time, torch
sys.path.insert(0, ".")
from semanticlens_amd import _native as N
DEV="cuda:0"
def wall(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/reps*1e6
for T,H,D,B in ((256,16,72,256),(256,16,72,64),(197,12,64,256),(50,12,64,256),(77,8,64,1024),(729,16,72,48)):
    qkv = torch.randn(B*T, 3*H*D, device=DEV); sp = N.Split(B*T, H*D, DEV)
    print(f"T={T} H={H} D={D} B={B}: {wall(lambda: N.attention(qkv, B, T, H, D, False, out_split=sp, bf16x3=True)):.1f} us", flush=True)
