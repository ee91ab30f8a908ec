"""The three matrix products of a linear layer on the hand-written MFMA GEMM kernels.

``nn.Linear`` is the reference model's only compute layer (reference min_DDP.py:44-45); its
forward and the two backward products are

    nt(x, w, b)   y  = x w^T + b    x [M, K], w [N, K]            (forward)
    nn(g, w)      dx = g w          g [M, N], w [N, K]            (input gradient)
    tn(g, x)      dw = g^T x        g [M, N], x [M, K]            (weight gradient)

Every fused autograd node of the package (MLP, q|k|v projection + attention, lm-head + loss) and
the plain linear op route their GEMMs through these three functions.  ``impl="native"`` (the
default everywhere) runs ``csrc/kernels/gemm_bf16.hip`` / ``gemm_f32.hip``; a shape or dtype those
kernels do not take (extents that are not multiples of the 16-byte vector width, fp16) falls back to
torch.  ``impl="library"`` sends the product to the vendor GEMM behind torch on purpose: it exists for
A/B measurements (``bench.py --linear-impl library``), not as a code path of the models.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .. import _ext
from .grad_sink import weight_grad_buffer

_NAMES = {torch.float32: "float32", torch.bfloat16: "bfloat16"}


def native_ok(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, a_kmajor: bool, b_kmajor: bool) -> bool:
    name = _NAMES.get(a.dtype)
    if name is None or a.dtype != b.dtype or not (a.is_cuda and b.is_cuda) or not _ext.have_C():
        return False
    if not (a.is_contiguous() and b.is_contiguous()) or min(M, N, K) <= 0:
        return False
    return bool(_ext.C().gemm_supported(M, N, K, a_kmajor, b_kmajor, name))


def nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, impl: str = "native") -> torch.Tensor:
    """x [M, K] @ w [N, K]^T (+ bias)."""
    if impl == "native" and native_ok(x, w, x.shape[0], w.shape[0], x.shape[1], True, True) and w.shape[0] > 16 and x.shape[1] > 1:
        return _ext.C().gemm(x, w, True, True, bias)
    return F.linear(x, w, bias)


def nn(g: torch.Tensor, w: torch.Tensor, impl: str = "native") -> torch.Tensor:
    """g [M, N] @ w [N, K]."""
    if impl == "native" and native_ok(g, w, g.shape[0], w.shape[1], g.shape[1], True, False):
        return _ext.C().gemm(g, w, True, False)
    return g @ w


def tn(g: torch.Tensor, x: torch.Tensor, impl: str = "native", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """g [M, N]^T @ x [M, K]: the whole row count is the contraction (split over the chip when the output has few tiles).
    ``out``: a contiguous [N, K] tensor to write into (a gradient-bucket slice, see ``grad_sink.py``)."""
    if impl == "native" and native_ok(g, x, g.shape[1], x.shape[1], g.shape[0], False, False):
        return _ext.C().gemm(g, x, False, False, None, 0, out)
    if out is not None:
        return torch.mm(g.t(), x, out=out)
    return g.t() @ x


def weight_grad(g: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, impl: str = "native") -> torch.Tensor:
    """dW = g^T x for ``weight``, written straight into the parameter's gradient-bucket slice when the
    data-parallel wrapper offers one."""
    out = weight_grad_buffer(weight)
    if out is not None and (out.shape != (g.shape[1], x.shape[1]) or out.dtype != g.dtype):
        out = None
    return tn(g, x, impl, out)
