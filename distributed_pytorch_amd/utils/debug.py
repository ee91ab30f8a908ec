"""Debug switches for the stream / bucket machinery.

``DIST_DEBUG_SYNC=1``   every collective is followed by a device synchronisation, which
                        serialises the compute and communication streams (a race between them
                        disappears under it, which localises the bug).
``check_in_sync(ddp)``  all ranks compare a parameter checksum; a diverged replica is the
                        symptom of a missed fence or a bucket consumed before its all-reduce.
"""
from __future__ import annotations

import os

import torch

from ..parallel import process_group as pg


def debug_sync_enabled() -> bool:
    return os.environ.get("DIST_DEBUG_SYNC", "0") == "1"


def check_in_sync(model: torch.nn.Module, rtol: float = 0.0) -> bool:
    with torch.no_grad():
        c = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum()
    lo, hi = c.clone(), c.clone()
    pg.all_reduce(lo, "min")
    pg.all_reduce(hi, "max")
    return bool((hi - lo).abs() <= rtol * hi.abs())


def assert_in_sync(model: torch.nn.Module, step: int = -1) -> None:
    if not check_in_sync(model):
        raise RuntimeError(f"model replicas diverged across ranks (step {step}); rerun with DIST_DEBUG_SYNC=1")
