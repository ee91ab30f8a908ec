"""Times the fused lm-head loss + in-place gradient kernel on [N, 50304] bf16 logits (50257 real columns)."""
import sys, torch
from distributed_pytorch_amd import _ext
C = _ext.C()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
big = torch.randn(N, 50304, device="cuda").bfloat16()
tgt = torch.randint(0, 50257, (N,), device="cuda")
def timeit(fn, n=5):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e-3
t = timeit(lambda: C.xent_fused(big, tgt, 50257, -100, False))
print(f"N={N}: {t*1e3:.3f} ms, {2*big.numel()*2/t/1e12:.2f} TB/s")
