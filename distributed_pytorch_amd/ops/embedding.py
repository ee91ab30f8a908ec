"""Token + position embedding as one op: ``x[b, t, :] = wte[idx[b, t], :] + wpe[t, :]``.

The reference model has no embedding (reference min_DDP.py:41-49); this op is the first layer of the
GPT-2 benchmark configuration.  On gfx950 with bf16 tables it is one forward kernel (the stock path is a
gather followed by a broadcast add over the whole [B, T, C] tensor) and two backward kernels that add
rows up in a fixed order -- positions over the batch, tokens over their stable-sorted occurrences -- so
the gradients are reproducible without atomics.  Other devices / dtypes run the stock operators.

Token ids must lie in ``[0, V)``: the kernels clamp instead of faulting; ``DIST_CHECK_TARGETS=1`` makes the
op raise ``IndexError`` like the stock operator does (one host synchronisation per call).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from .. import _ext


class _TokenPositionEmbedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, wte, wpe):
        T = idx.shape[-1]
        out = _ext.C().embedding_forward(idx, wte, wpe, T)
        ctx.save_for_backward(idx)
        ctx.dims = (T, wte.shape[0], wpe.shape[0])
        return out.view(*idx.shape, wte.shape[1])

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        T, V, P = ctx.dims
        g2 = g.reshape(-1, g.shape[-1])
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        tok_sorted, perm = torch.sort(idx.reshape(-1), stable=True)
        dwte, dwpe = _ext.C().embedding_backward(g2, tok_sorted, perm, T, V, P)
        return None, dwte, dwpe


def native_embedding_supported(idx: torch.Tensor, wte: torch.Tensor, wpe: torch.Tensor) -> bool:
    if not (idx.is_cuda and idx.dim() >= 1 and _ext.have_C()):
        return False
    return bool(_ext.C().embedding_supported(idx, wte, wpe, idx.shape[-1]))


def token_position_embedding(idx: torch.Tensor, wte: torch.Tensor, wpe: torch.Tensor) -> torch.Tensor:
    """idx: int64 [..., T]; wte: [V, C]; wpe: [P >= T, C] -> [..., T, C]."""
    if native_embedding_supported(idx, wte, wpe):
        if os.environ.get("DIST_CHECK_TARGETS", "0") == "1":
            lo, hi = int(idx.min()), int(idx.max())
            if lo < 0 or hi >= wte.shape[0]:
                raise IndexError(f"index out of range in token_position_embedding: [{lo}, {hi}] for a table of {wte.shape[0]} rows")
        return _TokenPositionEmbedding.apply(idx, wte, wpe)
    pos = torch.arange(idx.shape[-1], device=idx.device)
    return F.embedding(idx, wte) + F.embedding(pos, wpe)
