"""Input-gradient GEMM of an MLP's second layer at GPT-2 shape: plain native kernel, native kernel with the
activation-backward epilogue (ReLU' / GELU', with and without bias column sums), and the vendor GEMM."""
import sys, torch
from distributed_pytorch_amd import _ext
from distributed_pytorch_amd.ops.linear import ACT_GELU, ACT_RELU
C = _ext.C()
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
out_f, hid = 768, 3072
dy = torch.randn(tokens, out_f, device="cuda").bfloat16()
w = (torch.randn(out_f, hid, device="cuda") / 28).bfloat16()
aux = torch.randn(tokens, hid, device="cuda").bfloat16()
fl = 2.0 * tokens * out_f * hid
cases = {
    "vendor dy @ w": lambda: dy @ w,
    "native plain": lambda: C.gemm(dy, w, True, False),
    "native relu'": lambda: C.dgrad_act_backward(dy, w, aux, ACT_RELU, False),
    "native relu' + colsum": lambda: C.dgrad_act_backward(dy, w, aux, ACT_RELU, True),
    "native gelu'": lambda: C.dgrad_act_backward(dy, w, aux, ACT_GELU, False),
    "native gelu' + colsum": lambda: C.dgrad_act_backward(dy, w, aux, ACT_GELU, True),
}
for rnd in range(3):  # round-robin: clocks drift over the first seconds of load
    for name, fn in cases.items():
        t = timeit(fn)
        print(f"round {rnd}  {name:24s} {t*1e3:7.0f} us  {fl/t/1e9:6.0f} TF", flush=True)
