"""Causal self-attention on a packed q|k|v projection.

``causal_attention_qkvpacked(qkv, n_head)`` takes the [B, T, 3*C] output of the fused
query/key/value projection as it comes out of the GEMM and returns [B, T, C] ready for the
output projection: the kernels read the three operands with their native row stride and write
heads back interleaved, so no head split / transpose / reshape copies exist in either
direction, and the backward writes dq|dk|dv straight into one [B, T, 3*C] tensor (the
gradient of the projection output).

On gfx950 the work runs on hand-written flash-attention kernels (bf16 MFMA, online softmax,
recompute-based backward without atomics) for head size 64.  The kernels work on 128-row
blocks: a sequence length that is not a multiple of 128 is zero-padded at the end up to the next
block (padded keys lie in every real query's future, so the causal mask already hides them, and
padded queries receive a zero output gradient, so they add nothing to dK / dV); the padded rows
are cut off again on the way out.  Other head sizes, dtypes and CPU tensors use torch's
scaled-dot-product attention.

The reference contains no attention (its model is two linear layers, reference
min_DDP.py:41-49); this op serves the GPT-2 benchmark configuration.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from .. import _ext
from . import gemm


class _FlashQKV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, n_head, scale):
        C = _ext.C()
        o, lse = C.attention_forward(qkv, n_head, scale)
        ctx.save_for_backward(qkv, o, lse)
        ctx.n_head, ctx.scale = n_head, scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse = ctx.saved_tensors
        d_o = d_o if d_o.is_contiguous() else d_o.contiguous()
        return _ext.C().attention_backward(d_o, qkv, o, lse, ctx.n_head, ctx.scale)[0], None, None


class _ProjFlashQKV(torch.autograd.Function):
    """q|k|v projection + attention as one node: the attention backward kernels sum their dq / dk / dv
    tiles over the rows on the way out, which is the projection's bias gradient, so the [tokens, 3*C]
    gradient is not re-read by a column-sum pass."""

    @staticmethod
    def forward(ctx, x2, w, b, B, T, n_head, scale, impl):
        qkv = gemm.nt(x2, w, b, impl).view(B, T, w.shape[0])
        o, lse = _ext.C().attention_forward(qkv, n_head, scale)
        ctx.save_for_backward(x2, w, qkv, o, lse)
        ctx.n_head, ctx.scale, ctx.impl = n_head, scale, impl
        return o

    @staticmethod
    def backward(ctx, d_o):
        C = _ext.C()
        x2, w, qkv, o, lse = ctx.saved_tensors
        d_o = d_o if d_o.is_contiguous() else d_o.contiguous()
        dqkv, db = C.attention_backward(d_o, qkv, o, lse, ctx.n_head, ctx.scale, True)
        g = dqkv.view(-1, dqkv.shape[-1])
        dx = gemm.nn(g, w, ctx.impl) if ctx.needs_input_grad[0] else None
        return dx, gemm.weight_grad(g, x2, w, ctx.impl), db, None, None, None, None, None


_BLOCK_ROWS = 128  # rows per query / key block of the kernels


def native_attention_supported(qkv: torch.Tensor, n_head: int, pad_ok: bool = False) -> bool:
    """True when the kernels take ``qkv`` as it is; with ``pad_ok`` also when they take it after the
    sequence has been padded to a whole number of 128-row blocks."""
    if not (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and (pad_ok or qkv.is_contiguous())):
        return False
    if not _ext.have_C():
        return False
    _, T, C3 = qkv.shape
    if T <= 0 or C3 % (3 * n_head):
        return False
    if pad_ok:
        T += (-T) % _BLOCK_ROWS
    return bool(_ext.C().attention_supported(T, C3 // 3 // n_head, C3))


def projected_causal_attention(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], n_head: int,
                               softmax_scale: Optional[float] = None, impl: str = "native") -> Optional[torch.Tensor]:
    """``attention(x @ weight.T + bias)`` for x [B, T, C_in] and a [3*C, C_in] q|k|v projection, as one
    autograd node on the native kernels.  Returns None when the shape / dtype is not covered (the
    caller then runs the projection and ``causal_attention_qkvpacked`` separately)."""
    if bias is None or x.dim() != 3 or not (x.is_cuda and x.dtype == weight.dtype == torch.bfloat16) or not _ext.have_C():
        return None
    B, T, _ = x.shape
    C3 = weight.shape[0]
    if C3 % (3 * n_head) or not _ext.C().attention_supported(T, C3 // 3 // n_head, C3) or (B * T) % 32:
        return None
    hd = C3 // 3 // n_head
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(hd)
    x2 = x.reshape(B * T, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    return _ProjFlashQKV.apply(x2, weight, bias, B, T, n_head, scale, impl)


def causal_attention_qkvpacked(qkv: torch.Tensor, n_head: int, softmax_scale: Optional[float] = None,
                               impl: str = "auto") -> torch.Tensor:
    """qkv: [B, T, 3*C] (q | k | v along the last dimension) -> [B, T, C].

    impl: "auto" (native kernels when the shape qualifies), "native" (raise otherwise), "sdpa".
    """
    B, T, C3 = qkv.shape
    Cd = C3 // 3
    hd = Cd // n_head
    scale = float(softmax_scale) if softmax_scale is not None else 1.0 / math.sqrt(hd)
    if impl != "sdpa":
        if torch.is_autocast_enabled() and qkv.is_cuda and qkv.dtype != torch.bfloat16:
            qkv = qkv.to(torch.get_autocast_dtype("cuda"))
        if native_attention_supported(qkv, n_head):
            return _FlashQKV.apply(qkv, n_head, scale)
        if native_attention_supported(qkv, n_head, pad_ok=True):
            padded = F.pad(qkv, (0, 0, 0, (-T) % _BLOCK_ROWS))  # contiguous [B, T rounded up, 3*C], zero rows at the end
            y = _FlashQKV.apply(padded, n_head, scale)
            return y if padded.shape[1] == T else y[:, :T].contiguous()
        if impl == "native":
            raise RuntimeError(f"native attention does not cover qkv {tuple(qkv.shape)} {qkv.dtype} on {qkv.device} "
                               f"with {n_head} heads (needs bf16 and head size 64)")
    q, k, v = (t.view(B, T, n_head, hd).transpose(1, 2) for t in qkv.split(Cd, dim=2))
    y = F.scaled_dot_product_attention(q, k, v, is_causal=True, scale=scale)
    return y.transpose(1, 2).reshape(B, T, Cd)
