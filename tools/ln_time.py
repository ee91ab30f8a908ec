"""Times the fused residual-add + layer norm forward and its backward on [rows, 768] bf16 (GPT-2 small's shape)."""
import sys, torch
from distributed_pytorch_amd import _ext
C = _ext.C()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
cols = 768
x = torch.randn(rows, cols, device="cuda").bfloat16()
r = torch.randn_like(x)
w = torch.ones(cols, device="cuda").bfloat16(); b = torch.zeros(cols, device="cuda").bfloat16()
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e-3
y, mean, rstd, s = C.layernorm_forward(x, w, b, 1e-5, r, None)
nb = x.numel() * 2
t_f = timeit(lambda: C.layernorm_forward(x, w, b, 1e-5, r, None))
dy = torch.randn_like(x); dres = torch.randn_like(x)
t_b = timeit(lambda: C.layernorm_backward(dy, s, w, mean, rstd, dres, False))
print(f"rows {rows}: add+LN forward {t_f*1e6:.0f} us ({4*nb/t_f/1e12:.2f} TB/s over 2 reads + 2 writes), backward {t_b*1e6:.0f} us ({4*nb/t_b/1e12:.2f} TB/s over 3 reads + 1 write)")
