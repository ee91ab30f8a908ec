"""Batch normalisation of channels-last activations on the gfx950 kernels, with the ReLU and the residual add
of a ResNet block folded in.

Not part of the reference's two-layer model (reference min_DDP.py:41-49); it serves the ResNet-50 data-parallel
configuration of the benchmark set, where the stock channels-last batch-norm kernels were 50 of the 80 ms of a
training step.  ``BatchNorm2d`` subclasses ``torch.nn.BatchNorm2d`` -- same parameters, buffers and state_dict --
and adds two optional arguments to ``forward``: ``y = relu(bn(x) + residual)``.  The native path needs a
channels-last (or 2-D) GPU tensor; anything else goes through ``torch.nn.functional.batch_norm``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _ext

_DTYPES = {torch.float32: "float32", torch.bfloat16: "bfloat16", torch.float16: "float16"}


def _rows(t: torch.Tensor) -> torch.Tensor:
    """[N, C, H, W] in channels-last memory -> the same memory as [N*H*W, C] (no copy); 2-D tensors pass through."""
    if t.dim() == 2:
        return t
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])


def _like(rows: torch.Tensor, shape) -> torch.Tensor:
    """Inverse of ``_rows`` for a tensor of the given (2-D or NCHW) shape."""
    if len(shape) == 2:
        return rows
    n, c, h, w = shape
    return rows.view(n, h, w, c).permute(0, 3, 1, 2)


def _native_ok(x: torch.Tensor, residual: Optional[torch.Tensor]) -> bool:
    if not x.is_cuda or x.dtype not in _DTYPES or not _ext.have_C():
        return False
    if x.dim() == 4:
        if not x.is_contiguous(memory_format=torch.channels_last):
            return False
        rows, chans = x.shape[0] * x.shape[2] * x.shape[3], x.shape[1]
    elif x.dim() == 2:
        if not x.is_contiguous():
            return False
        rows, chans = x.shape
    else:
        return False
    if residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype or residual.stride() != x.stride()):
        return False
    return rows > 1 and bool(_ext.C().batchnorm_supported(rows, chans, _DTYPES[x.dtype]))


class _BatchNormAct(torch.autograd.Function):
    """y = relu(batch_norm(x) + residual) with batch statistics (training mode).  Saves x and the statistics; for
    the ReLU mask the forward's per-channel coefficients (the mask is recomputed from x) or, when a residual was
    added before the ReLU, the output y, which the next layer keeps alive anyway."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum: float, eps: float, residual, relu: bool):
        x2 = _rows(x)
        r2 = _rows(residual) if residual is not None else None
        y2, mean, rstd, coef = _ext.C().batchnorm_forward(x2, weight, bias, running_mean, running_var, momentum, eps, r2, relu)
        mask_from_x = relu and residual is None
        ctx.save_for_backward(x2, y2 if relu and not mask_from_x else None, coef if mask_from_x else None, weight, mean, rstd)
        ctx.relu = relu
        ctx.has_residual = residual is not None
        ctx.shape = tuple(x.shape)
        return _like(y2, ctx.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, y2, coef, weight, mean, rstd = ctx.saved_tensors
        shape = ctx.shape
        if len(shape) == 4 and not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dy2 = _rows(dy) if len(shape) == 4 else dy.contiguous()
        want_dres = ctx.has_residual and ctx.needs_input_grad[7] and ctx.relu
        dx2, dgamma, dbeta, dres2 = _ext.C().batchnorm_backward(dy2, x2, y2, coef, weight, mean, rstd, ctx.relu, want_dres)
        dres = None
        if ctx.has_residual and ctx.needs_input_grad[7]:
            dres = _like(dres2, shape) if ctx.relu else dy  # without the ReLU the residual's gradient is dy itself
        dw = dgamma if weight is not None and ctx.needs_input_grad[1] else None
        db = dbeta if ctx.needs_input_grad[2] else None
        return _like(dx2, shape), dw, db, None, None, None, None, dres, None


def batch_norm_act(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], running_mean: Optional[torch.Tensor],
                   running_var: Optional[torch.Tensor], training: bool, momentum: float, eps: float,
                   residual: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """``relu(batch_norm(x) + residual)``; ``residual`` and ``relu`` optional.  Native for channels-last GPU tensors in
    training mode (and, without autograd, in evaluation mode); ``F.batch_norm`` plus separate add / ReLU otherwise."""
    if _native_ok(x, residual):
        if training or running_mean is None:
            return _BatchNormAct.apply(x, weight, bias, running_mean if training else None, running_var if training else None,
                                       float(momentum), float(eps), residual, bool(relu))
        needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, weight, bias, residual))
        if not needs_grad:
            scale = torch.rsqrt(running_var.float() + eps)
            if weight is not None:
                scale = scale * weight.float()
            shift = -running_mean.float() * scale
            if bias is not None:
                shift = shift + bias.float()
            y2 = _ext.C().batchnorm_apply(_rows(x), scale.contiguous(), shift.contiguous(), _rows(residual) if residual is not None else None,
                                          bool(relu))
            return _like(y2, tuple(x.shape))
    y = F.batch_norm(x, running_mean, running_var, weight, bias, training or running_mean is None, momentum, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class BatchNorm2d(nn.BatchNorm2d):
    """``torch.nn.BatchNorm2d`` with ``forward(x, residual=None, relu=False)`` = ``relu(bn(x) + residual)``."""

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:  # type: ignore[override]
        self._check_input_dim(x)
        factor = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:  # cumulative moving average
                factor = 1.0 / float(self.num_batches_tracked)
        use_batch_stats = self.training or (self.running_mean is None and self.running_var is None)
        rm = self.running_mean if (not self.training or self.track_running_stats) else None
        rv = self.running_var if (not self.training or self.track_running_stats) else None
        return batch_norm_act(x, self.weight, self.bias, rm, rv, use_batch_stats, factor, self.eps, residual, relu)
