"""Layer normalisation on the gfx950 kernel (one wave per row, row held in registers).

Not part of the reference's two-layer model; it is a hot elementwise/reduction op of the
GPT-2 configuration that the framework is benchmarked on, so it gets a native kernel with
the same fp32 statistics as ``torch.nn.functional.layer_norm``.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _ext


def _supported(x: torch.Tensor, cols: int, need_grad: bool) -> bool:
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return False
    return 1 <= cols <= 8192  # any width: register-resident rows where they fit, a workgroup per row otherwise


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, weight, bias, eps: float):
        y, mean, rstd, _ = _ext.C().layernorm_forward(x2, weight, bias, eps)
        ctx.save_for_backward(x2, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight, mean, rstd = ctx.saved_tensors
        dx, dg, db, _ = _ext.C().layernorm_backward(dy.contiguous(), x2, weight, mean, rstd)
        return dx, dg, db, None


class _KeepLayerNorm(torch.autograd.Function):
    """(x, LN(x)) for an ``x`` that also continues as the residual stream: handing the stream back through the
    node lets the backward kernel add the stream's gradient to the layer-norm gradient in its own pass, where
    autograd would otherwise join the two with a separate add over the whole tensor."""

    @staticmethod
    def forward(ctx, x2, weight, bias, eps: float):
        y, mean, rstd, _ = _ext.C().layernorm_forward(x2, weight, bias, eps)
        ctx.save_for_backward(x2, weight, mean, rstd)
        return x2.view_as(x2), y

    @staticmethod
    def backward(ctx, ds, dy):
        x2, weight, mean, rstd = ctx.saved_tensors
        if dy is None:
            return ds, None, None, None
        dres = ds.contiguous() if ds is not None else None
        dx, dg, db, _ = _ext.C().layernorm_backward(dy.contiguous(), x2, weight, mean, rstd, dres, False)
        return dx, dg, db, None


class _AddLayerNorm(torch.autograd.Function):
    """(s, y) = (x + r [+ r_bias], LN(s)) in one pass; the backward folds the gradient that reaches
    ``s`` from its other consumers (the residual stream) into the layer-norm gradient in one pass
    too.  ``x`` and ``r`` receive the same gradient tensor; ``r_bias`` (the bias of the layer that
    produced ``r``, added here instead of in that layer's GEMM) receives its column sums, which the
    same kernel accumulates next to the layer-norm parameter gradients."""

    @staticmethod
    def forward(ctx, x2, r2, r_bias, weight, bias, eps: float):
        y, mean, rstd, s = _ext.C().layernorm_forward(x2, weight, bias, eps, r2, r_bias)
        ctx.save_for_backward(s, weight, mean, rstd)
        ctx.has_r_bias = r_bias is not None
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        s, weight, mean, rstd = ctx.saved_tensors
        if dy is None:  # only the sum was used downstream
            drb = ds.sum(0) if ctx.has_r_bias else None
            return ds, ds, drb, None, None, None
        dres = ds.contiguous() if ds is not None else None
        dx, dg, db, drb = _ext.C().layernorm_backward(dy.contiguous(), s, weight, mean, rstd, dres, ctx.has_r_bias)
        return dx, dx, (drb if ctx.has_r_bias else None), dg, db, None


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Normalises over the last dimension with affine ``weight`` / ``bias``."""
    cols = x.shape[-1]
    need_grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)
    if bias is None or not _supported(x, cols, need_grad) or weight.dtype != x.dtype:
        return F.layer_norm(x, (cols,), weight, bias, eps)
    x2 = x.reshape(-1, cols)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    return _LayerNorm.apply(x2, weight.contiguous(), bias.contiguous(), eps).reshape(x.shape)


def add_layer_norm(x: torch.Tensor, r, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, r_bias=None):
    """Returns ``(s, y)`` with ``s = x + r (+ r_bias)`` and ``y = layer_norm(s)``; ``r`` may be
    ``None`` (then ``s`` is ``x``).  One kernel on the GPU in each direction instead of add + layer
    norm; ``r_bias`` lets the layer that produced ``r`` skip its own bias add and bias-gradient pass."""
    cols = x.shape[-1]
    if r is None:
        need_grad = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)
        if bias is None or not x.requires_grad or not _supported(x, cols, need_grad) or weight.dtype != x.dtype:
            return x, layer_norm(x, weight, bias, eps)
        x2 = x.reshape(-1, cols)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        s, y = _KeepLayerNorm.apply(x2, weight.contiguous(), bias.contiguous(), eps)
        return s.reshape(x.shape), y.reshape(x.shape)
    if r_bias is not None and (not x.is_cuda or cols > 8192 or r_bias.dtype != x.dtype):
        r = r + r_bias
        r_bias = None
    need_grad = torch.is_grad_enabled() and (x.requires_grad or r.requires_grad or weight.requires_grad)
    if (bias is None or not _supported(x, cols, need_grad) or weight.dtype != x.dtype or r.dtype != x.dtype
            or r.shape != x.shape or r.device != x.device):
        s = x + (r if r_bias is None else r + r_bias)
        return s, F.layer_norm(s, (cols,), weight, bias, eps)
    x2, r2 = x.reshape(-1, cols), r.reshape(-1, cols)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    r2 = r2 if r2.is_contiguous() else r2.contiguous()
    s, y = _AddLayerNorm.apply(x2, r2, None if r_bias is None else r_bias.contiguous(), weight.contiguous(), bias.contiguous(), eps)
    return s.reshape(x.shape), y.reshape(x.shape)


class LayerNorm(nn.LayerNorm):
    """``nn.LayerNorm`` over the last dimension (same parameters and ``state_dict`` keys)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if len(self.normalized_shape) != 1 or self.weight is None:
            return super().forward(x)
        return layer_norm(x, self.weight, self.bias, self.eps)

    def forward_add(self, x: torch.Tensor, r, r_bias=None):
        """``(s, self(s))`` with ``s = x + r (+ r_bias)``, fused; ``r`` may be ``None``."""
        if len(self.normalized_shape) != 1 or self.weight is None:
            s = x if r is None else x + (r if r_bias is None else r + r_bias)
            return s, super().forward(s)
        return add_layer_norm(x, r, self.weight, self.bias, self.eps, r_bias)
