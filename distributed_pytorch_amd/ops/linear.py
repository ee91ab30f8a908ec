"""Linear layer with fused bias + activation on the hand-written gfx950 kernels.

``y = act(x W^T + b)`` is what ``nn.Linear`` (plus an activation) computes for the
reference model (reference min_DDP.py:41-49).  On the GPU the forward and all three
backward products run on our own kernels (``csrc/kernels``): MFMA GEMMs for GEMM-shaped
layers, dedicated streaming kernels for the reference's degenerate K = 1 and N = 4 layers.
On CPU tensors the same module computes with stock torch ops, which is what the CPU
oracle runs.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _ext
from .grad_sink import weight_grad_buffer

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
_ACT_CODES = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "gelu": ACT_GELU}


def _act_code(act) -> int:
    if isinstance(act, int):
        return act
    try:
        return _ACT_CODES[act if act is None else act.lower()]
    except KeyError:
        raise ValueError(f"unknown activation '{act}' (none | relu | gelu)") from None


def _torch_act(y: torch.Tensor, code: int) -> torch.Tensor:
    if code == ACT_RELU:
        return F.relu(y)
    if code == ACT_GELU:
        return F.gelu(y)
    return y


def native_route(M: int, N: int, K: int, dtype: torch.dtype) -> str:
    """Kernel family used for a [M,K] x [N,K]^T product: outer | skinny | mfma | unsupported."""
    name = {torch.float32: "float32", torch.bfloat16: "bfloat16", torch.float16: "float16"}.get(dtype)
    if name is None:
        return "unsupported"
    return _ext.C().linear_route(M, N, K, name)


class _LinearBiasAct(torch.autograd.Function):
    """2-D x only; the module wrapper flattens leading dimensions."""

    @staticmethod
    def forward(ctx, x, weight, bias, act: int):
        C = _ext.C()
        need_pre = act == ACT_GELU and (x.requires_grad or weight.requires_grad)
        K = x.shape[1]
        if K == 1 or weight.shape[0] <= 16:
            # The streaming kernels fuse the activation only on the K = 1 path.
            fused_act = act if K == 1 and act != ACT_GELU else ACT_NONE
            y, _ = C.linear_forward(x, weight, bias, fused_act, False)
            pre = None
            if fused_act != act:
                pre = y
                y = _torch_act(y, act)
        else:
            y, pre = C.linear_forward(x, weight, bias, act, need_pre)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y if act == ACT_RELU else (pre if act == ACT_GELU else None))
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.C()
        x, weight, saved = ctx.saved_tensors
        dy = dy.contiguous()
        need_dx = ctx.needs_input_grad[0]
        db = None
        if ctx.act != ACT_NONE:
            # g = dy * act'(.) and the bias gradient come out of one pass over dy.
            g, db = C.act_backward(dy, saved, ctx.act, ctx.has_bias)
            dx, dw, _ = C.linear_backward(g, x, weight, need_dx, False, weight_grad_buffer(weight))
        else:
            dx, dw, db = C.linear_backward(dy, x, weight, need_dx, ctx.has_bias, weight_grad_buffer(weight))
        return dx, dw, (db if ctx.has_bias else None), None


class _LibraryGemmLinear(torch.autograd.Function):
    """A/B comparison path only (``impl="library"``): the three GEMMs on the vendor library behind
    torch.mm / addmm, everything around them (activation backward fused with the bias gradient in one
    pass over dy, deterministic two-stage column sum) on our kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, act: int):
        pre = F.linear(x, weight, bias)
        y = _torch_act(pre, act)
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, pre if act == ACT_GELU else (y if act == ACT_RELU else None))
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.C()
        x, weight, saved = ctx.saved_tensors
        dy = dy.contiguous()
        db = None
        if ctx.act != ACT_NONE:
            g, db = C.act_backward(dy, saved, ctx.act, ctx.has_bias)
        else:
            g = dy
            if ctx.has_bias:
                db = C.colsum(dy)
        dx = g @ weight if ctx.needs_input_grad[0] else None
        dw = g.t() @ x
        return dx, dw, db, None


def library_linear_bias_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None) -> torch.Tensor:
    """``act(x @ weight.T + bias)`` with vendor-library GEMMs and native elementwise/reduction kernels."""
    code = _act_code(act)
    if not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        return _torch_act(F.linear(x, weight, bias), code)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    y = _LibraryGemmLinear.apply(x2, weight, bias, code)
    return y.reshape(*lead, weight.shape[0])


def linear_bias_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act=None,
                    allow_library_gemm: bool = False) -> torch.Tensor:
    """``act(x @ weight.T + bias)`` for x of shape ``[..., K]``.

    GPU tensors of float32 / bfloat16 / float16 run on the native kernels, whatever their extents
    (odd sizes take the element-wise tail path of the 128x128 MFMA kernel).  What is left for
    "unsupported" is a dtype outside those three, mixed x / weight dtypes or a leading dimension
    beyond 4 Mi elements: that raises, unless ``allow_library_gemm`` asks for the vendor GEMM
    behind torch instead.
    """
    code = _act_code(act)
    if not x.is_cuda:
        return _torch_act(F.linear(x, weight, bias), code)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    route = native_route(x2.shape[0], weight.shape[0], x2.shape[1], x2.dtype) if x2.dtype == weight.dtype else "unsupported"
    if route == "unsupported":
        if not allow_library_gemm:
            raise RuntimeError(f"linear_bias_act: no native kernel for x{tuple(x2.shape)} w{tuple(weight.shape)} {x2.dtype}")
        return _torch_act(F.linear(x, weight, bias), code)
    w = weight if weight.is_contiguous() else weight.contiguous()
    y = _LinearBiasAct.apply(x2, w, bias, code)
    return y.reshape(*lead, weight.shape[0])


class Linear(nn.Linear):
    """``nn.Linear`` (same parameters, same ``state_dict`` keys) with an optional fused
    activation.  ``impl="native"`` (default) computes everything on the hand-written kernels;
    ``impl="library"`` sends the GEMMs to the vendor library for A/B measurements."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, act=None, device=None, dtype=None,
                 impl: str = "native"):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        if impl not in ("native", "library"):
            raise ValueError(f"impl must be native | library, got {impl}")
        self.act = act
        self.impl = impl

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.impl == "library" and x.is_cuda:
            return library_linear_bias_act(x, self.weight, self.bias, self.act)
        return linear_bias_act(x, self.weight, self.bias, self.act)

    def extra_repr(self) -> str:
        return super().extra_repr() + (f", act={self.act}" if self.act else "") + (f", impl={self.impl}" if self.impl != "native" else "")
