"""Softmax cross-entropy fused with argmax accuracy.

One kernel pass over the logits gives the loss *and* the per-sample "prediction correct"
flags that the reference computes separately with ``nn.CrossEntropyLoss`` and
``torch.argmax(y_hat, 1).eq(y)`` (reference min_DDP.py:75,100,106); aggregate metrics
(loss sum, correct count, sample count) stay on the device so a training loop can log
without a host synchronisation per step.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _ext
from . import gemm


class XentResult(NamedTuple):
    loss: torch.Tensor      # scalar (mean / sum) or [N] (reduction='none')
    correct: torch.Tensor   # bool [N]
    metrics: torch.Tensor   # float32 [4]: loss sum, correct count, counted rows, mean loss


class _SoftmaxXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, reduction: str, ignore_index: int, n_classes: int, inplace_backward: bool):
        C = _ext.C()
        row_loss, row_lse, correct, metrics = C.xent_forward(logits, targets, n_classes, ignore_index)
        ctx.save_for_backward(logits, targets, row_lse, metrics)
        ctx.cfg = (reduction, ignore_index, n_classes, inplace_backward)
        ctx.mark_non_differentiable(correct, metrics)
        if reduction == "mean":
            loss = metrics[3].clone()  # float32 scalar whatever the logits dtype
        elif reduction == "sum":
            loss = metrics[0].clone()
        else:
            loss = row_loss
        return loss, correct, metrics

    @staticmethod
    def backward(ctx, dloss, _dcorrect, _dmetrics):
        C = _ext.C()
        logits, targets, row_lse, metrics = ctx.saved_tensors
        reduction, ignore_index, n_classes, inplace = ctx.cfg
        if reduction == "none":
            # Per-row upstream gradients: scale rows afterwards (rare path).
            one = torch.ones((), dtype=torch.float32, device=logits.device)
            d = C.xent_backward(logits, targets, row_lse, one, metrics, False, n_classes, ignore_index, False)
            return d * dloss.to(d.dtype).unsqueeze(1), None, None, None, None, None
        g = dloss.detach().to(torch.float32).reshape(1).contiguous()
        d = C.xent_backward(logits, targets, row_lse, g, metrics, reduction == "mean", n_classes, ignore_index, inplace)
        return d, None, None, None, None, None


class _ProjectionXent(torch.autograd.Function):
    """loss(x @ W^T): the vocabulary projection and its loss as ONE autograd node, so that the
    gradient of the logits can be written by the same kernel launch that computes the loss (a 16-bit
    row of up to 57344 columns stays in the workgroup's registers between the two passes; longer or
    fp32 rows are read a second time, out of cache).  The node's backward is then only
    the two GEMMs; an upstream gradient other than 1 scales their (small) outputs instead of the
    [tokens, vocabulary] tensor."""

    @staticmethod
    def forward(ctx, x2, weight, targets, reduction: str, ignore_index: int, n_classes: int, impl: str):
        C = _ext.C()
        logits = gemm.nt(x2, weight, None, impl)
        correct, metrics = C.xent_fused(logits, targets, n_classes, ignore_index, reduction == "mean")
        ctx.save_for_backward(x2, weight, logits)  # `logits` now holds d loss / d logits for a unit upstream gradient
        ctx.impl = impl
        ctx.mark_non_differentiable(correct, metrics)
        loss = (metrics[3] if reduction == "mean" else metrics[0]).clone()
        return loss, correct, metrics

    @staticmethod
    def backward(ctx, dloss, _dcorrect, _dmetrics):
        x2, weight, dlogits = ctx.saved_tensors
        g = dloss.detach().to(dlogits.dtype)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = gemm.nn(dlogits, weight, ctx.impl).mul_(g)
        if ctx.needs_input_grad[1]:
            # [vocabulary, K] with the whole token count as contraction
            dw = gemm.weight_grad(dlogits, x2, weight, ctx.impl).mul_(g)
        return dx, dw, None, None, None, None, None


def projection_cross_entropy(x: torch.Tensor, weight: torch.Tensor, targets: torch.Tensor, reduction: str = "mean",
                             ignore_index: int = -100, n_classes: Optional[int] = None, impl: str = "native",
                             check_targets: Optional[bool] = None) -> XentResult:
    """Cross-entropy of ``x @ weight.T`` (``x``: [..., K], ``weight``: [C, K]) against ``targets``.

    On the GPU the logits exist only inside the op: they are written by the GEMM, turned into
    their own gradient in place by the fused loss kernel and consumed by the two backward GEMMs.
    ``n_classes`` < C marks trailing rows of ``weight`` as padding.  reduction: mean | sum.
    """
    if reduction not in ("mean", "sum"):
        raise ValueError(f"reduction must be mean | sum, got {reduction}")
    x2 = x.reshape(-1, x.shape[-1])
    t1 = targets.reshape(-1)
    nc = weight.shape[0] if n_classes is None else int(n_classes)
    if _want_target_check(check_targets):
        check_targets_in_range(t1, nc, ignore_index)
    fused_ok = (x2.is_cuda and x2.dtype == weight.dtype and x2.dtype in (torch.float32, torch.bfloat16, torch.float16)
                and nc > 64 and _ext.have_C())
    if not fused_ok:
        return softmax_cross_entropy(F.linear(x2, weight), t1, reduction, ignore_index, nc, inplace_backward=True, check_targets=False)
    loss, correct, metrics = _ProjectionXent.apply(x2.contiguous(), weight, t1.contiguous(), reduction, ignore_index, nc, impl)
    return XentResult(loss, correct, metrics)


def check_targets_in_range(targets: torch.Tensor, n_classes: int, ignore_index: int) -> None:
    """Raises ``IndexError`` when a target is neither a class index nor ``ignore_index`` (what
    ``nn.CrossEntropyLoss`` reports through a device-side assert).  Costs a host synchronisation, so the
    loss functions only run it when asked (``check_targets=True`` or ``DIST_CHECK_TARGETS=1``); without
    the check the kernels treat such a row like an ignored one."""
    bad = (targets != ignore_index) & ((targets < 0) | (targets >= n_classes))
    if bool(bad.any()):
        v = int(targets[bad][0])
        raise IndexError(f"Target {v} is out of bounds (classes 0..{n_classes - 1}, ignore_index {ignore_index})")


def _want_target_check(flag: Optional[bool]) -> bool:
    import os

    return bool(flag) if flag is not None else os.environ.get("DIST_CHECK_TARGETS", "0") == "1"


def softmax_cross_entropy(logits: torch.Tensor, targets: torch.Tensor, reduction: str = "mean", ignore_index: int = -100,
                          n_classes: Optional[int] = None, inplace_backward: bool = False,
                          check_targets: Optional[bool] = None) -> XentResult:
    """Cross-entropy of ``logits[N, C]`` against class indices ``targets[N]``.

    ``n_classes`` < C marks trailing columns as padding (a vocabulary padded for aligned
    rows): they take no part in the softmax and get zero gradient.  ``inplace_backward``
    lets the backward overwrite ``logits`` with their gradient, which saves an ``N x C``
    allocation when the logits are not needed again.
    """
    if reduction not in ("mean", "sum", "none"):
        raise ValueError(f"reduction must be mean | sum | none, got {reduction}")
    if logits.dim() != 2:
        raise ValueError("softmax_cross_entropy expects logits of shape [N, C]")
    C_ = logits.shape[1]
    nc = C_ if n_classes is None else int(n_classes)
    if _want_target_check(check_targets):
        check_targets_in_range(targets, nc, ignore_index)
    if not logits.is_cuda:
        lg = logits[:, :nc] if nc != C_ else logits
        row = F.cross_entropy(lg.float(), targets, reduction="none", ignore_index=ignore_index)
        valid = targets != ignore_index
        correct = (lg.argmax(1) == targets) & valid
        nvalid = valid.sum().clamp(min=1)
        metrics = torch.stack([row.sum(), correct.sum().float(), valid.sum().float(), row.sum() / nvalid]).detach()
        loss = row.sum() / nvalid if reduction == "mean" else (row.sum() if reduction == "sum" else row)
        return XentResult(loss, correct, metrics)
    if not logits.is_contiguous():
        logits = logits.contiguous()
    loss, correct, metrics = _SoftmaxXent.apply(logits, targets.contiguous(), reduction, ignore_index, nc, inplace_backward)
    return XentResult(loss, correct, metrics)


class SoftmaxCrossEntropy(nn.Module):
    """Criterion with ``nn.CrossEntropyLoss`` call signature; after a call,
    ``last_correct`` / ``last_metrics`` hold the accuracy by-products of the same pass."""

    def __init__(self, reduction: str = "mean", ignore_index: int = -100, n_classes: Optional[int] = None,
                 inplace_backward: bool = False, check_targets: Optional[bool] = None):
        super().__init__()
        self.check_targets = check_targets
        self.reduction = reduction
        self.ignore_index = ignore_index
        self.n_classes = n_classes
        self.inplace_backward = inplace_backward
        self.last_correct: Optional[torch.Tensor] = None
        self.last_metrics: Optional[torch.Tensor] = None

    def forward(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        if logits.dim() > 2:
            logits = logits.reshape(-1, logits.shape[-1])
            targets = targets.reshape(-1)
        r = softmax_cross_entropy(logits, targets, self.reduction, self.ignore_index, self.n_classes, self.inplace_backward,
                                  self.check_targets)
        self.last_correct, self.last_metrics = r.correct, r.metrics
        return r.loss
