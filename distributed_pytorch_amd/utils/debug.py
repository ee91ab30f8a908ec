"""Debug switches for the stream / bucket machinery.

``DIST_DEBUG_SYNC=1``
    * every facade collective is followed by a device synchronisation;
    * every bucket all-reduce the reducer queues on the communication stream is followed by a host
      wait for that stream (``RcclComm::set_debug_sync``), so compute and communication never run
      concurrently: a race between the two streams disappears under the flag, which localises it;
    * ``assert_in_sync`` compares a parameter checksum across ranks (without the flag it returns at
      once, so a training loop can call it at every logging window at no cost).
``check_in_sync(model)``  the comparison itself, unconditional: a diverged replica is the symptom of a
    missed fence or of a bucket consumed before its all-reduce finished.
"""
from __future__ import annotations

import torch

from ..parallel import process_group as pg


def debug_sync_enabled() -> bool:
    return pg.debug_sync()


def check_in_sync(model: torch.nn.Module, rtol: float = 0.0) -> bool:
    if hasattr(model, "check_replicas_in_sync"):
        return model.check_replicas_in_sync(rtol)
    with torch.no_grad():
        c = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum()
    lo, hi = c.clone(), c.clone()
    pg.all_reduce(lo, "min")
    pg.all_reduce(hi, "max")
    return bool((hi - lo).abs() <= rtol * hi.abs())


def assert_in_sync(model: torch.nn.Module, step: int = -1) -> None:
    """No-op unless ``DIST_DEBUG_SYNC=1``; raises when the replicas' parameter checksums differ."""
    if not debug_sync_enabled():
        return
    if not check_in_sync(model):
        raise RuntimeError(f"model replicas diverged across ranks (step {step})")
