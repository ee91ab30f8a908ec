"""Times the fc forward (393216 x 768 -> 3072) with its epilogue variants, next to a plain fill of the same bytes:
is the two-output GELU epilogue bound by its stores?"""
import sys
import torch
from distributed_pytorch_amd import _ext

C = _ext.C()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
N, K = 3072, 768


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
b = torch.zeros(N, device="cuda").bfloat16()
fl = 2.0 * M * N * K
for name, fn in (
    ("plain", lambda: C.linear_forward(x, w, None, 0, False)),
    ("bias", lambda: C.linear_forward(x, w, b, 0, False)),
    ("bias+gelu, one output", lambda: C.linear_forward(x, w, b, 2, False)),
    ("bias+gelu, two outputs", lambda: C.linear_forward(x, w, b, 2, True)),
):
    t = timeit(fn)
    print(f"{name:28s} {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s", flush=True)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
out2 = torch.empty_like(out)
t = timeit(lambda: out.zero_())
print(f"fill {out.numel() * 2 / 1e9:.2f} GB: {t:8.1f} us  {out.numel() * 2 / t / 1e6:.2f} TB/s")
t = timeit(lambda: (out.zero_(), out2.zero_()))
print(f"two fills: {t:8.1f} us  {2 * out.numel() * 2 / t / 1e6:.2f} TB/s")
t = timeit(lambda: out2.copy_(out))
print(f"copy: {t:8.1f} us  {2 * out.numel() * 2 / t / 1e6:.2f} TB/s")
