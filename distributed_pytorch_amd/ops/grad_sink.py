"""Where an operator may write a parameter's gradient.

Under the data-parallel wrapper every parameter owns a slice of a flat gradient bucket
(``parallel/ddp.py``).  An operator that computes a weight gradient with its own kernel asks
``weight_grad_buffer(param)`` for that slice, lets the kernel write into it and returns it as the
gradient: autograd adopts the tensor as ``param.grad`` without a copy, so the bucket is complete the
moment the kernel has run and the reducer has nothing to pack.  ``None`` means "allocate as usual"
(no wrapper, a gradient already exists and must be accumulated into, second use of the parameter in
one backward pass, bucket dtype differs from the parameter's).
"""
from __future__ import annotations

from typing import Optional

import torch

_ATTR = "_dpa_grad_sink"


def register(param: torch.Tensor, provider) -> None:
    """``provider()`` returns the tensor to write into, or None."""
    setattr(param, _ATTR, provider)


def unregister(param: torch.Tensor) -> None:
    if hasattr(param, _ATTR):
        delattr(param, _ATTR)


def weight_grad_buffer(param: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    provider = getattr(param, _ATTR, None) if param is not None else None
    return provider() if provider is not None else None
