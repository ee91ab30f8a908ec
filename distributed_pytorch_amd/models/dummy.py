"""The reference's toy workload: an index-valued dataset and a two-layer linear model.

``DummyDataset`` reproduces reference min_DDP.py:27-38 value for value (inputs equal their
index so shard membership is visible in logs; labels from ``Generator().manual_seed(0)``).
``DummyModel`` has the layers, parameter names and (absence of) activation of reference
min_DDP.py:41-49, so its ``state_dict`` is interchangeable; with ``native=True`` the layers
run on the gfx950 kernels when the tensors are on the GPU.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.utils.data import Dataset

from ..ops.linear import Linear
from ..ops.rank1_mlp import rank1_mlp, rank1_mlp_supported


class DummyDataset(Dataset):
    def __init__(self, length: int, n_classes: int):
        self.len = length
        gen = torch.Generator().manual_seed(0)
        self.data = torch.arange(0, length, dtype=torch.float32).unsqueeze(-1)
        self.labels = torch.randint(0, n_classes, (length,), generator=gen)

    def __getitem__(self, idx):
        return self.data[idx], self.labels[idx]

    def __len__(self):
        return self.len


class DummyModel(nn.Module):
    """``lin2(lin1(x))`` with no nonlinearity in between (as in the reference).

    With ``native=True`` on the GPU and ``in_dim == 1`` (the reference's configuration) both layers run
    as one fused node that keeps the hidden activation in registers (``ops/rank1_mlp.py``);
    ``fused=False`` keeps them as two separately executed linear layers."""

    def __init__(self, in_dim: int, hidden_dim: int, n_classes: int, native: bool = True, act=None, fused: bool = True):
        super().__init__()
        layer = Linear if native else nn.Linear
        self.lin1 = layer(in_dim, hidden_dim) if not (native and act) else Linear(in_dim, hidden_dim, act=act)
        self.lin2 = layer(hidden_dim, n_classes)
        self.native = bool(native)
        self.fused = bool(fused)
        self.act = act

    def forward(self, x):
        if self.native and self.fused and rank1_mlp_supported(x, self.lin1.weight, self.lin2.weight, self.act):
            return rank1_mlp(x, self.lin1.weight, self.lin1.bias, self.lin2.weight, self.lin2.bias, self.act)
        return self.lin2(self.lin1(x))
