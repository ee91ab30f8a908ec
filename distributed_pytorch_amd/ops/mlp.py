"""Two-layer MLP ``proj(act(fc(x)))`` as one autograd node.

A transformer MLP (``c_fc`` -> GELU -> ``c_proj``) is not in the reference's two-layer toy model
(reference min_DDP.py:41-49, which has no nonlinearity at all); it is the largest block of the
GPT-2 benchmark configuration.  Treating the pair of layers as one node lets the backward fuse
across the layer boundary: the input-gradient GEMM of ``proj`` applies the activation derivative
of ``fc`` in its epilogue and accumulates ``fc``'s bias gradient on the way out, so the
[tokens, 4*C] gradient is written once and never re-read by an elementwise pass.

    forward   pre, h = x W_fc^T + b_fc, act(.)   MFMA GEMM, both outputs from one epilogue (GELU; GEMM + elementwise
                                                   pass for other activations)
              y   = h W_proj^T (+ b_proj)  MFMA GEMM
    backward  g, db_fc = (dy W_proj) * act'(pre), colsum(.)   MFMA GEMM, fused epilogue
              dW_proj  = dy^T h                                split-K MFMA GEMM
              dx       = g W_fc                                MFMA GEMM
              dW_fc    = g^T x                                 split-K MFMA GEMM

``impl="library"`` sends the three plain products to the vendor GEMM (A/B measurements only).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .. import _ext
from . import gemm
from .linear import ACT_GELU, ACT_RELU, _act_code, _torch_act, linear_bias_act


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, w_fc, b_fc, w_proj, b_proj, act: int, impl: str):
        if act == ACT_GELU:
            # one GEMM writes the activation and the pre-activation from its epilogue
            h, pre = _ext.C().linear_forward(x2, w_fc, b_fc, act, True)
        else:
            pre = gemm.nt(x2, w_fc, b_fc, impl)
            h = _torch_act(pre, act)
        y = gemm.nt(h, w_proj, b_proj, impl)
        ctx.act = act
        ctx.impl = impl
        ctx.has_b_proj = b_proj is not None
        ctx.save_for_backward(x2, w_fc, w_proj, pre if act == ACT_GELU else h, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.C()
        x2, w_fc, w_proj, aux, h = ctx.saved_tensors
        dy = dy.contiguous()
        g, db_fc = C.dgrad_act_backward(dy, w_proj, aux, ctx.act, True)
        dw_proj = gemm.weight_grad(dy, h, w_proj, ctx.impl)
        db_proj = C.colsum(dy) if ctx.has_b_proj else None
        dx = gemm.nn(g, w_fc, ctx.impl) if ctx.needs_input_grad[0] else None
        dw_fc = gemm.weight_grad(g, x2, w_fc, ctx.impl)
        return dx, dw_fc, db_fc, dw_proj, db_proj, None, None


def fused_mlp_supported(x: torch.Tensor, w_fc: torch.Tensor, b_fc, w_proj: torch.Tensor, act) -> bool:
    if not (x.is_cuda and x.dtype == torch.bfloat16 and w_fc.dtype == torch.bfloat16 and w_proj.dtype == torch.bfloat16):
        return False
    if b_fc is None or _act_code(act) not in (ACT_GELU, ACT_RELU) or not _ext.have_C():
        return False
    tokens = x.numel() // x.shape[-1]
    hidden, out = w_fc.shape[0], w_proj.shape[0]
    # the fused epilogue lives in the 256x256-tile kernel: whole tiles only
    return bool(_ext.C().dgrad_act_backward_supported(tokens, hidden, out)) and w_proj.is_contiguous() and w_fc.is_contiguous()


def fused_mlp(x: torch.Tensor, w_fc: torch.Tensor, b_fc: Optional[torch.Tensor], w_proj: torch.Tensor,
              b_proj: Optional[torch.Tensor], act="gelu", impl: str = "native") -> torch.Tensor:
    """``act(x @ w_fc.T + b_fc) @ w_proj.T (+ b_proj)`` for x of shape ``[..., K]``.

    Falls back to the two separate linear ops when the shapes do not fill whole 256x256 tiles or the
    tensors are not bf16 on the GPU.
    """
    code = _act_code(act)
    if not fused_mlp_supported(x, w_fc, b_fc, w_proj, code):
        if x.is_cuda:
            return linear_bias_act(linear_bias_act(x, w_fc, b_fc, code), w_proj, b_proj, None)
        return F.linear(_torch_act(F.linear(x, w_fc, b_fc), code), w_proj, b_proj)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    y = _FusedMLP.apply(x2, w_fc, b_fc, w_proj, b_proj, code, impl)
    return y.reshape(*lead, w_proj.shape[0])
