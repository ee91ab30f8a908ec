"""Vendor-GEMM solution table (PyTorch TunableOp) for the plain library GEMMs.

TunableOp's own timings suggest large wins over hipBLASLt's default heuristic on the GPT-2 shapes
(0.167 ms against 0.204 ms for the 65536x2304x768 projection), but they do not reproduce in the
training step: with the recorded table the same call measures 0.251 ms against 0.262 ms and the
whole step gains 0.3 %, so the benchmark leaves this OFF by default (``bench.py --tuned-gemms``
opts in).  ``enable_tuned_gemms()`` switches TunableOp on in
look-up-only mode with the table that ships in ``distributed_pytorch_amd/tuning/`` -- no tuning
happens at run time, shapes that are not in the table use the library default, and a table recorded
for another library build is ignored by torch's validators.  ``record_tuned_gemms(path)`` is the
producing side: it makes torch time every candidate the first time a shape is seen and write the
winners to ``path`` (run a few training steps under it, then copy the file into the package).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

_TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tuning", "gfx950_gemm.csv")


def default_table() -> str:
    return _TABLE


def enable_tuned_gemms(path: Optional[str] = None) -> bool:
    """Look-up-only TunableOp with the shipped table.  Returns False (and changes nothing) when
    there is no GPU, no table, or this torch build has no TunableOp."""
    path = path or _TABLE
    if not torch.cuda.is_available() or not os.path.isfile(path) or not hasattr(torch.cuda, "tunable"):
        return False
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(False)
    if hasattr(t, "record_untuned_enable"):
        t.record_untuned_enable(False)
    t.set_filename(path, insert_device_ordinal=False)
    if hasattr(t, "write_file_on_exit"):
        t.write_file_on_exit(False)
    return bool(t.read_file(path))


def record_tuned_gemms(path: str, max_duration_ms: int = 200, max_iterations: int = 100) -> None:
    """Tuning mode: every new GEMM shape is timed over all library solutions; winners go to ``path``."""
    t = torch.cuda.tunable
    t.enable(True)
    t.tuning_enable(True)
    t.set_max_tuning_duration(max_duration_ms)
    t.set_max_tuning_iterations(max_iterations)
    t.set_filename(path, insert_device_ordinal=False)
    if hasattr(t, "write_file_on_exit"):
        t.write_file_on_exit(True)  # newer builds append results as they are found


def disable() -> None:
    if hasattr(torch.cuda, "tunable"):
        torch.cuda.tunable.enable(False)
