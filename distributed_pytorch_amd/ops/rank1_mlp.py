"""``Linear(1, H) -> [ReLU] -> Linear(H, N)`` as one autograd node that never stores the hidden layer.

This is the reference model's whole forward pass (reference min_DDP.py:41-49: ``lin2(lin1(x))`` with
``in_dim = 1`` and four classes).  Each hidden value is one FMA of the row's input scalar, so both
directions recompute it in registers (``csrc/kernels/rank1_mlp.hip``) instead of writing and re-reading
an ``[M, H]`` tensor: at M = 65536, H = 4096 that removes about 5 GB of HBM traffic per training step.
Tensors on the CPU -- and inputs that themselves need a gradient -- take the layer-by-layer route.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .. import _ext
from .linear import ACT_NONE, ACT_RELU, _act_code


class _Rank1MLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act: int):
        C = _ext.C()
        y = C.rank1_mlp_forward(x, w1, b1, w2, b2, act)
        ctx.act = act
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.save_for_backward(x, w1, b1, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.C()
        x, w1, b1, w2 = ctx.saved_tensors
        dw1, db1, dw2, db2 = C.rank1_mlp_backward(dy.contiguous(), x, w1, b1, w2, ctx.act)
        return None, dw1, (db1 if ctx.has_b1 else None), dw2, (db2 if ctx.has_b2 else None), None


def rank1_mlp_supported(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, act) -> bool:
    return (_ext.have_C() and x.is_cuda and w1.is_cuda and w2.is_cuda and x.dim() == 2 and x.shape[1] == 1 and w1.shape[1] == 1 and w2.shape[0] <= 16 and not x.requires_grad
            and x.dtype == w1.dtype == w2.dtype and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
            and _act_code(act) in (ACT_NONE, ACT_RELU))


def rank1_mlp(x: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor],
              act=None) -> torch.Tensor:
    """``act(x @ w1.T + b1) @ w2.T + b2`` for x of shape ``[M, 1]``."""
    code = _act_code(act)
    if not rank1_mlp_supported(x, w1, w2, act):
        h = F.linear(x, w1, b1)
        if code == ACT_RELU:
            h = F.relu(h)
        elif code != ACT_NONE:
            h = F.gelu(h)
        return F.linear(h, w2, b2)
    xc = x if x.is_contiguous() else x.contiguous()
    return _Rank1MLP.apply(xc, w1.contiguous(), b1, w2.contiguous(), b2, code)
