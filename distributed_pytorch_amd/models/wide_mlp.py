"""Wide MLP: ``depth`` hidden layers of ``hidden`` units with fused bias + ReLU.

The data-parallel stress configuration "8 x 8192": eight 8192 x 8192 layers are 537 M
parameters (2.1 GB of fp32 gradients per step), so it exercises the MFMA GEMMs with their
fused epilogue and the bucketed all-reduce at once.  It scales up the reference's
two-layer toy model (reference min_DDP.py:41-49); unlike it, the layers have a
nonlinearity, which is what the fused epilogue is for.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..ops.linear import Linear


class WideMLP(nn.Module):
    def __init__(self, in_dim: int = 8192, hidden: int = 8192, depth: int = 8, n_classes: int = 1024, act: str = "relu",
                 linear_impl: str = "native"):
        super().__init__()
        dims = [in_dim] + [hidden] * depth
        self.layers = nn.ModuleList([Linear(dims[i], dims[i + 1], act=act, impl=linear_impl) for i in range(depth)])
        self.head = Linear(hidden, n_classes, impl=linear_impl)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for layer in self.layers:
            x = layer(x)
        return self.head(x)

    def flops_per_sample(self) -> float:
        return 6.0 * sum(p.numel() for n, p in self.named_parameters() if n.endswith("weight"))
