"""Forward / backward check and timing of the native attention against torch SDPA."""
import sys, math, torch
import torch.nn.functional as F
from distributed_pytorch_amd import _ext
C = _ext.C()
B, T, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 256, 3)
do_bwd = len(sys.argv) > 4 and sys.argv[4] == "bwd"
torch.manual_seed(0)
Cd = H * 64
qkv = torch.randn(B, T, 3 * Cd, device="cuda").bfloat16()
scale = 1.0 / math.sqrt(64)
o, lse = C.attention_forward(qkv, H, scale)
q, k, v = (t.view(B, T, H, 64).transpose(1, 2).float() for t in qkv.split(Cd, dim=2))
q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
ref = F.scaled_dot_product_attention(q, k, v, is_causal=True)
ref_o = ref.transpose(1, 2).reshape(B, T, Cd)
s = (q @ k.transpose(-1, -2)) * scale
mask = torch.ones(T, T, device="cuda", dtype=torch.bool).tril()
ref_lse = torch.logsumexp(s.masked_fill(~mask, float("-inf")), -1)
print("fwd max err", (o.float() - ref_o).abs().max().item(), "lse err", (lse - ref_lse).abs().max().item())
if do_bwd:
    go = torch.randn_like(o)
    dqkv, _ = C.attention_backward(go, qkv, o, lse, H, scale)
    ref_o.backward(go.float())
    rg = torch.cat([t.grad.transpose(1, 2).reshape(B, T, Cd) for t in (q, k, v)], dim=2)
    for name, a, b in zip("qkv", dqkv.float().split(Cd, 2), rg.split(Cd, 2)):
        print("d" + name, "max err", (a - b).abs().max().item(), "ref max", b.abs().max().item())
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
if B * T >= 32768:
    qb, kb, vb = (t.view(B, T, H, 64).transpose(1, 2) for t in qkv.split(Cd, dim=2))
    t_native = timeit(lambda: C.attention_forward(qkv, H, scale))
    t_sdpa = timeit(lambda: F.scaled_dot_product_attention(qb, kb, vb, is_causal=True))
    fl = 4 * B * H * T * T * 64 / 2
    print(f"fwd native {t_native*1e3:.1f} us ({fl/t_native/1e9:.0f} TF)  sdpa {t_sdpa*1e3:.1f} us ({fl/t_sdpa/1e9:.0f} TF)")
    if do_bwd:
        go = torch.randn_like(o)
        t_nb = timeit(lambda: C.attention_backward(go, qkv, o, lse, H, scale))
        qg, kg, vg = (t.detach().clone().requires_grad_() for t in (qb, kb, vb))
        out = F.scaled_dot_product_attention(qg, kg, vg, is_causal=True)
        gg = torch.randn_like(out)
        t_sb = timeit(lambda: torch.autograd.grad(out, (qg, kg, vg), gg, retain_graph=True))
        print(f"bwd native {t_nb*1e3:.1f} us ({2.5*fl/t_nb/1e9:.0f} TF)  sdpa {t_sb*1e3:.1f} us ({2.5*fl/t_sb/1e9:.0f} TF)")
