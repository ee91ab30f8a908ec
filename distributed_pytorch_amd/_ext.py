"""Loading of the two native modules.

``_host`` (g++: TCP store, bucket planner) is needed everywhere and is built on first
import if the shared object is missing.  ``_C`` (hipcc, gfx950: HIP kernels, RCCL
communicator, reducer) is needed whenever a GPU is used; a missing ``_C`` on a machine
that has a GPU is a hard error, never a silent fall back to stock PyTorch ops.
"""
from __future__ import annotations

import importlib
import os
import sys
import threading
from pathlib import Path

_ROOT = Path(__file__).resolve().parent.parent
_lock = threading.Lock()
_host = None
_C = None
_C_error: Exception | None = None


def _build(host_only: bool):
    sys.path.insert(0, str(_ROOT))
    try:
        import build as _build_mod  # the repository-level build.py

        _build_mod.build_all(host_only=host_only)
    finally:
        sys.path.pop(0)


def host():
    """The torch-free native module (TCP store + bucket planner)."""
    global _host
    if _host is None:
        with _lock:
            if _host is None:
                try:
                    _host = importlib.import_module("distributed_pytorch_amd._host")
                except ImportError:
                    if os.environ.get("DIST_NO_AUTOBUILD") == "1":
                        raise
                    _build(host_only=True)
                    importlib.invalidate_caches()
                    _host = importlib.import_module("distributed_pytorch_amd._host")
    return _host


def gpu_available() -> bool:
    import torch

    return torch.cuda.is_available()


def C(required: bool = True):
    """The HIP/RCCL module.  ``required=False`` returns ``None`` instead of raising."""
    global _C, _C_error
    if _C is None and _C_error is None:
        with _lock:
            if _C is None and _C_error is None:
                try:
                    import torch  # noqa: F401  (loads libamdhip64 / librccl the module links to)

                    _C = importlib.import_module("distributed_pytorch_amd._C")
                except Exception as e:  # ImportError or a loader error
                    _C_error = e
    if _C is None and required:
        raise RuntimeError(
            "distributed_pytorch_amd._C (HIP kernels / RCCL communicator) is not built or failed to "
            f"load: {_C_error!r}. Run `python build.py` in {_ROOT}."
        )
    return _C


def have_C() -> bool:
    return C(required=False) is not None
