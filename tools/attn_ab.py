"""Same-box A/B of the two attention forward kernels: runs tools/attn_check.py-style timing for each in a subprocess
(the kernel is chosen once per process by DPA_ATTN_FWD)."""
import os, subprocess, sys
shapes = [(64, 1024, 12), (32, 1024, 12), (8, 1024, 12), (16, 2048, 16), (4, 4096, 16), (96, 512, 12)]
if len(sys.argv) > 1 and sys.argv[1] == "threshold":  # where does the second kernel start to pay?  (work items = pairs of 128-query blocks of two heads)
    shapes = [(12, 1024, 12), (16, 1024, 12), (22, 1024, 12), (24, 1024, 12), (2, 4096, 16), (4, 2048, 16), (6, 2048, 16), (24, 512, 12), (48, 512, 12)]
code = r'''
import sys, math, torch
from distributed_pytorch_amd import _ext
C = _ext.C()
B, T, H = (int(v) for v in sys.argv[1:4])
qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
scale = 0.125
def timeit(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
t = timeit(lambda: C.attention_forward(qkv, H, scale))
fl = 4 * B * H * T * T * 64 / 2
print(f"{t*1e3:.1f} us {fl/t/1e9:.0f} TF")
'''
for B, T, H in shapes:
    row = []
    for kind in ("classic", "wide", "tile32", "classic", "wide", "tile32"):
        env = dict(os.environ, DPA_ATTN_FWD=kind)
        out = subprocess.run([sys.executable, "-c", code, str(B), str(T), str(H)], env=env, capture_output=True, text=True, timeout=120)
        row.append(f"{kind}: {out.stdout.strip() or out.stderr.strip()[-200:]}")
    print(f"B{B} T{T} H{H}:  " + "   ".join(row), flush=True)
