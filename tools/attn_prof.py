"""Runs the native attention forward + backward a few times (profiling target)."""
import math, sys, torch
from distributed_pytorch_amd import _ext
C = _ext.C()
B, T, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 1024, 12)
torch.manual_seed(0)
qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
scale = 1.0 / math.sqrt(64)
for _ in range(3):
    o, lse = C.attention_forward(qkv, H, scale)
    d, _ = C.attention_backward(torch.ones_like(o), qkv, o, lse, H, scale)
torch.cuda.synchronize()
