"""Times the fused AdamW kernel on a Wide-MLP-sized parameter set (8 x 8192 x 8192 bf16 with fp32 master weights)."""
import torch
from distributed_pytorch_amd.ops import FusedAdamW
ps = [torch.nn.Parameter(torch.randn(8192, 8192, device="cuda").bfloat16()) for _ in range(8)]
for p in ps: p.grad = torch.randn_like(p)
opt = FusedAdamW(ps, lr=1e-4, weight_decay=0.01, master_weights=True)
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e-3
t = timeit(opt.step)
n = sum(p.numel() for p in ps)
print(f"{n/1e6:.0f} M parameters: {t*1e3:.3f} ms per step, {n*(2+2+4+4+4+2+4+4+4)/t/1e12:.2f} TB/s")
