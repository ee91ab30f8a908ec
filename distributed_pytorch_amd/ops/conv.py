"""Convolutions as GEMMs over channels-last rows: 1x1 (a plain product) and 3x3 stride 1 (an implicit product).

Not part of the reference's two-layer model (reference min_DDP.py:41-49); it serves the ResNet-50 configuration of
the benchmark set.  In channels-last memory a 1x1, stride-1 convolution IS ``y[R, Cout] = x[R, Cin] @ W[Cout, Cin]^T``
with R = batch * height * width, i.e. exactly the linear layer the hand-written MFMA GEMMs serve (forward NT,
input gradient NN, weight gradient split-K TN) -- about half of the convolution FLOPs of a bottleneck ResNet.
``PointwiseConv2d`` subclasses ``torch.nn.Conv2d`` (same parameters and state_dict); anything the GEMM path does not
cover (other layouts, strides, groups, CPU tensors) goes through the stock convolution, and so do layers with
fewer than ``MIN_CHANNELS`` input or output channels: measured at batch 256, bf16, forward + backward
(``tools/bench_pointwise_conv.py``), the GEMM path takes 131-134 us against 237-282 us for the 1024<->256 and
2048<->512 layers of stages 3 and 4, 302-305 against 318-320 us for 512<->128, but 603-613 against 481-486 us for
256<->64 -- a 256-wide tile is mostly padding there.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _ext
from .linear import linear_bias_act, native_route


MIN_CHANNELS = 128
# 3x3 layers narrower than this stay on the stock convolution: the implicit-GEMM path computes on 256 x 256 output
# tiles, which a 64- or 128-channel layer fills to a quarter / a half.  Measured at batch 256, bf16, forward + backward
# (tools/bench_conv3x3.py): 64 ch 56x56 997 us against 592 us stock, 128 ch 28x28 511 against 419, but 256 ch 14x14
# 221 against 415 and 512 ch 7x7 307 against 497.
MIN_CHANNELS_3X3 = 256


class PointwiseConv2d(nn.Conv2d):
    def __init__(self, in_channels: int, out_channels: int, stride: int = 1, bias: bool = False, device=None, dtype=None):
        super().__init__(in_channels, out_channels, 1, stride=stride, bias=bias, device=device, dtype=dtype)

    def _gemm_ok(self, x: torch.Tensor) -> bool:
        if not (x.is_cuda and x.dim() == 4 and x.dtype == self.weight.dtype and x.is_contiguous(memory_format=torch.channels_last)):
            return False
        if self.stride != (1, 1) or self.padding != (0, 0) or self.dilation != (1, 1) or self.groups != 1:
            return False
        if min(self.in_channels, self.out_channels) < MIN_CHANNELS:
            return False
        rows = x.shape[0] * x.shape[2] * x.shape[3]
        return native_route(rows, self.out_channels, self.in_channels, x.dtype) == "mfma"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self._gemm_ok(x):
            return super().forward(x)
        n, c, h, w = x.shape
        x2 = x.permute(0, 2, 3, 1).reshape(-1, c)                # the same memory, [R, Cin]
        w2 = self.weight.reshape(self.out_channels, c)           # [Cout, Cin, 1, 1] is [Cout, Cin] in either memory format
        y2 = linear_bias_act(x2, w2, self.bias, None, allow_library_gemm=False)
        return y2.view(n, h, w, self.out_channels).permute(0, 3, 1, 2)


class _Conv3x3S1(torch.autograd.Function):
    """3x3, stride 1, padding 1, no bias, bf16, channels-last, as implicit GEMMs on the large-tile MFMA kernel.

    The input is copied once into a zero-haloed image X [B, H+2, W+2, C] (``conv_pad.hip``).  Forward: row m of the
    GEMM's A operand is output pixel (b, h, w) and its contraction index runs over (dh, dw, c) -- three runs of 3*C
    contiguous elements of X -- so the operand tiles are fetched by LDS-DMA straight from X with per-lane pixel
    offsets and a wave-uniform base that walks the taps; nothing like an im2col matrix exists in memory.  Input
    gradient: the same product over the haloed output gradient with the weights rotated by 180 degrees.  Weight
    gradient: one split-K product that contracts over ALL haloed pixels (halo rows are zero in both operands), whose
    B operand reads X shifted by the tap: dW[o][(dh, dw, c)] = sum_q dY[q][o] X[q + tap][c]."""

    @staticmethod
    def forward(ctx, x, weight):
        C = _ext.C()
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        xpad = C.conv3x3_pad(x.permute(0, 2, 3, 1))                     # [B, H, W, C] view of the same memory
        w2 = weight.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)          # channels-last weight: [Cout][(dh, dw, c)]
        y2 = C.conv3x3_forward(xpad, w2, B, H, W)
        ctx.save_for_backward(xpad, weight)
        ctx.geom = (B, H, W)
        return y2.view(B, H, W, Cout).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        C = _ext.C()
        xpad, weight = ctx.saved_tensors
        B, H, W = ctx.geom
        Cout, Cin = weight.shape[0], weight.shape[1]
        dypad = C.conv3x3_pad(dy.permute(0, 2, 3, 1).contiguous())
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # w_rot[c][(dh, dw, o)] = w[o][c][2 - dh][2 - dw]
            w_rot = weight.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous()
            dx = C.conv3x3_forward(dypad, w_rot, B, H, W).view(B, H, W, Cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = C.conv3x3_wgrad(dypad, xpad, B, H, W).view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        return dx, dw


class Conv3x3(nn.Conv2d):
    """``nn.Conv2d(cin, cout, 3, stride, padding=1, bias=False)`` (same parameters and state_dict).  Stride-1 layers
    whose channel counts are multiples of 64 run on the implicit-GEMM path above for bf16 channels-last GPU tensors;
    everything else (strided layers, other dtypes / layouts, CPU) goes through the stock convolution."""

    def __init__(self, in_channels: int, out_channels: int, stride: int = 1, device=None, dtype=None, min_channels: int = None):
        super().__init__(in_channels, out_channels, 3, stride=stride, padding=1, bias=False, device=device, dtype=dtype)
        self.min_channels = MIN_CHANNELS_3X3 if min_channels is None else int(min_channels)

    def _implicit_ok(self, x: torch.Tensor) -> bool:
        if not (_ext.have_C() and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and self.weight.dtype == torch.bfloat16):
            return False
        if self.stride != (1, 1) or self.padding != (1, 1) or self.dilation != (1, 1) or self.groups != 1 or self.bias is not None:
            return False
        if self.in_channels % 64 or self.out_channels % 64 or min(self.in_channels, self.out_channels) < self.min_channels:
            return False
        if not (x.is_contiguous(memory_format=torch.channels_last) and self.weight.is_contiguous(memory_format=torch.channels_last)):
            return False
        b, c, h, w = x.shape
        return (b * (h + 2) * (w + 2) + 2 * (w + 3) + 64) * max(self.in_channels, self.out_channels) * 2 < (1 << 32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self._implicit_ok(x):
            return super().forward(x)
        return _Conv3x3S1.apply(x, self.weight)
