"""Runs one GEMM shape a few times (for counter collection under rocprofv3)."""
import sys
import torch
from distributed_pytorch_amd import _ext
C = _ext.C()
M, N, K = (int(v) for v in sys.argv[1:4])
ak, bk = sys.argv[4] == "1", sys.argv[5] == "1"
which = sys.argv[6] if len(sys.argv) > 6 else "native"
A = torch.randn((M, K) if ak else (K, M), device="cuda").bfloat16()
B = torch.randn((N, K) if bk else (K, N), device="cuda").bfloat16()
for _ in range(5):
    if which == "native":
        C.gemm(A, B, ak, bk)
    else:
        torch.matmul(A if ak else A.t(), B.t() if bk else B)
torch.cuda.synchronize()
