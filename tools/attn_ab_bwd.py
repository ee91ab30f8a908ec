"""Same-box A/B of the attention backward kernels on 16x16x32 ("classic") and 32x32x16 ("wide") matrix instructions."""
import math, sys, torch
from distributed_pytorch_amd import _ext
C = _ext.C()
shapes = [(64, 1024, 12), (32, 1024, 12), (8, 1024, 12), (16, 2048, 16), (4, 4096, 16), (96, 512, 12)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
for B, T, H in shapes:
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
    o, lse = C.attention_forward(qkv, H, 0.125)
    go = torch.randn_like(o)
    fl = 2.5 * 4 * B * H * T * T * 64 / 2
    row = []
    for kind in ("classic", "wide", "classic", "wide"):
        t = timeit(lambda: C.attention_backward(go, qkv, o, lse, H, 0.125, False, kind))
        row.append(f"{kind}: {t*1e3:.1f} us {fl/t/1e9:.0f} TF")
    print(f"B{B} T{T} H{H}:  " + "   ".join(row), flush=True)
