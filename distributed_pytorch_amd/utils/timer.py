"""Step timing without host synchronisation inside the timed window, and profiler ranges.

The reference's only instrumentation is a ``print`` per iteration that forces several
device-to-host copies (reference min_DDP.py:110-116, 128-130).  ``StepTimer`` brackets a
window of steps with two HIP events and reads them once at the end; ``range`` emits a
roctx range (visible in ``rocprofv3 --marker-trace``) around forward / backward / optimizer.
"""
from __future__ import annotations

import contextlib
import time
from typing import List, Optional

import torch


class StepTimer:
    """Times windows of steps; warm-up windows are excluded by simply not starting the timer."""

    def __init__(self, device: Optional[torch.device] = None):
        self.use_events = torch.cuda.is_available() and (device is None or device.type == "cuda")
        self._start = None
        self._windows_ms: List[float] = []
        self._steps: List[int] = []

    def start(self) -> None:
        if self.use_events:
            self._start = torch.cuda.Event(enable_timing=True)
            self._start.record()
        else:
            self._start = time.perf_counter()

    def stop(self, n_steps: int) -> float:
        """Ends the window; this is the only point that waits for the device."""
        if self.use_events:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            end.synchronize()
            ms = self._start.elapsed_time(end)
        else:
            ms = (time.perf_counter() - self._start) * 1e3
        self._windows_ms.append(ms)
        self._steps.append(n_steps)
        return ms / max(n_steps, 1)

    def ms_per_step(self) -> float:
        return sum(self._windows_ms) / max(sum(self._steps), 1)

    def throughput(self, samples_per_step: int) -> float:
        return samples_per_step / (self.ms_per_step() * 1e-3)


_trace_backend = None  # (push, pop) once resolved


def _resolve_trace():
    """The extension's roctx hooks (the ones the reducer's own bucket ranges go through), or torch's."""
    global _trace_backend
    if _trace_backend is not None:
        return _trace_backend
    try:
        from .. import _ext

        C = _ext.C()
        if C.trace_enable(True):
            _trace_backend = (C.trace_push, C.trace_pop)
            return _trace_backend
    except Exception:
        pass
    if torch.cuda.is_available():
        _trace_backend = (torch.cuda.nvtx.range_push, torch.cuda.nvtx.range_pop)
    else:
        _trace_backend = (lambda _n: None, lambda: None)
    return _trace_backend


@contextlib.contextmanager
def range(name: str):  # noqa: A001 - deliberate: mirrors torch.cuda.nvtx.range
    """Profiler range around a phase of the step (forward / backward / optimizer); shows up in
    ``rocprofv3 --marker-trace`` together with the reducer's per-bucket launch ranges."""
    push, pop = _resolve_trace()
    push(name)
    try:
        yield
    finally:
        pop()


def allreduce_bus_bandwidth_gbps(nbytes: int, seconds: float, world_size: int) -> float:
    """Bus bandwidth of an all-reduce: algorithm bandwidth x 2(n-1)/n."""
    if seconds <= 0 or world_size < 1:
        return 0.0
    return nbytes / seconds * (2.0 * (world_size - 1) / world_size) / 1e9
