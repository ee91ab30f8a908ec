"""1x1 convolution as a GEMM over channels-last rows.

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

from .linear import linear_bias_act, native_route


MIN_CHANNELS = 128


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
