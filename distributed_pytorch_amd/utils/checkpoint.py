"""Checkpoint helpers that keep the reference's implicit layout.

The reference never saves anything, but its wrapping rule fixes the layout a user sees:
``state_dict()`` keys carry a ``module.`` prefix when the model was wrapped for more than one
rank and no prefix otherwise (reference distributed.py:112-115).  These helpers write on the
primary rank only, synchronise, and load on every rank onto that rank's device; ``strip`` /
``add`` convert between the two layouts so a checkpoint from an 8-GPU run loads on one GPU.
"""
from __future__ import annotations

import os
import tempfile
from typing import Any, Dict, Optional

import torch

from ..parallel import process_group as pg

PREFIX = "module."


def strip_module_prefix(state: Dict[str, Any]) -> Dict[str, Any]:
    return {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in state.items()}


def add_module_prefix(state: Dict[str, Any]) -> Dict[str, Any]:
    return {(k if k.startswith(PREFIX) else PREFIX + k): v for k, v in state.items()}


def _match_layout(state: Dict[str, Any], model: torch.nn.Module) -> Dict[str, Any]:
    want_prefix = any(k.startswith(PREFIX) for k in model.state_dict().keys())
    has_prefix = any(k.startswith(PREFIX) for k in state.keys())
    if want_prefix and not has_prefix:
        return add_module_prefix(state)
    if has_prefix and not want_prefix:
        return strip_module_prefix(state)
    return state


def save_checkpoint(path: str, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer] = None, step: int = 0,
                    extra: Optional[dict] = None) -> None:
    """Rank 0 writes atomically (temp file + rename); all ranks leave together.  An optimizer whose state is
    sharded over the ranks (``consolidated_state_dict``, collective) is saved in the unsharded layout, so the
    file loads on any world size and into a plain optimizer."""
    opt_state = None
    if optimizer is not None:
        gather = getattr(optimizer, "consolidated_state_dict", None)
        opt_state = gather() if gather is not None else (optimizer.state_dict() if pg.get_rank() == 0 else None)
    if pg.get_rank() == 0:
        payload = {
            "model": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "optimizer": opt_state,
            "step": int(step),
            "world_size": pg.get_world_size(),
            "extra": extra or {},
        }
        d = os.path.dirname(os.path.abspath(path))
        os.makedirs(d, exist_ok=True)
        fd, tmp = tempfile.mkstemp(dir=d, suffix=".tmp")
        os.close(fd)
        torch.save(payload, tmp)
        os.replace(tmp, path)
    pg.barrier()


def load_checkpoint(path: str, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer] = None, strict: bool = True) -> dict:
    """Every rank loads onto its own device; returns the metadata (step, world_size, extra)."""
    payload = torch.load(path, map_location=pg.get_device(), weights_only=True)
    model.load_state_dict(_match_layout(payload["model"], model), strict=strict)
    if optimizer is not None and payload.get("optimizer") is not None:
        optimizer.load_state_dict(payload["optimizer"])
    pg.barrier()
    return {"step": payload.get("step", 0), "world_size": payload.get("world_size", 1), "extra": payload.get("extra", {})}
