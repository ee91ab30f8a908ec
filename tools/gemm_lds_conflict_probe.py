"""Counter target: which part of the large-tile GEMM produces LDS bank conflicts.  Four launches in a fixed order
(read the per-dispatch counters in that order): plain NT, NT + bias + GELU epilogue (table look-ups), plain NN,
NN with the activation-backward epilogue (aux image + table look-ups).

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d out -- python3 tools/gemm_lds_conflict_probe.py
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from distributed_pytorch_amd import _ext


def main():
    C = _ext.C()
    M, K, N = 32768, 768, 3072
    dev = "cuda"
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    C.gemm(x, w, True, True)                                   # 1: plain NT
    h, pre = C.linear_forward(x, w, b, 2, True)                # 2: NT + bias + GELU, both outputs
    dy = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w2 = (torch.randn(K, N, device=dev) * 0.03).to(torch.bfloat16)   # c_proj weight [out=K, in=N]
    C.gemm(dy, w2, True, False)                                # 3: plain NN  (dy [M,K] @ w2 [K,N])
    C.dgrad_act_backward(dy, w2, pre, 2, True)                 # 4: NN + GELU' epilogue + column sums
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
