"""Weight-gradient GEMM (dW = dY^T X, contraction over tokens): native split-K kernel vs torch (hipBLASLt)."""
import sys, torch
from distributed_pytorch_amd import _ext
C = _ext.C()
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
for n_out, k_in in [(2304, 768), (768, 768), (3072, 768), (768, 3072), (4096, 1024), (1024, 4096), (50304, 768)]:
    g = torch.randn(tokens, n_out, device="cuda").bfloat16()
    x = torch.randn(tokens, k_in, device="cuda").bfloat16()
    ref = g.t() @ x
    out = C.gemm(g, x, False, False)
    err = (out.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
    fl = 2.0 * tokens * n_out * k_in
    tn, tl = timeit(lambda: C.gemm(g, x, False, False)), timeit(lambda: g.t() @ x)
    print(f"dW[{n_out},{k_in}] tokens {tokens}: rel err {err:.2e}  native {tn*1e3:.0f} us ({fl/tn/1e9:.0f} TF)  library {tl*1e3:.0f} us ({fl/tl/1e9:.0f} TF)", flush=True)
