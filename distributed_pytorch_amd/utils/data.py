"""Host -> device input pipeline: pinned staging buffers and a dedicated copy stream.

The reference moves every batch with a blocking pageable ``x.to(device)`` on the compute stream
(reference min_DDP.py:96).  ``DevicePrefetcher`` keeps the same iteration contract -- it yields
what the wrapped loader yields, with every tensor on the device -- but batch i+1 is copied while
batch i is being computed: tensors are staged in page-locked memory (reused ring of buffers, no
per-batch ``hipHostMalloc``), the copy is issued on its own HIP stream, and the compute stream
only waits on the copy's event.  On a CPU device it is a transparent pass-through.
"""
from __future__ import annotations

from typing import Any, Iterable, Iterator, List, Optional

import torch


def _map(obj: Any, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        out = [_map(o, fn) for o in obj]
        return type(obj)(out) if not hasattr(obj, "_fields") else type(obj)(*out)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    return obj


class DevicePrefetcher:
    """Iterates ``loader`` one batch ahead, delivering batches on ``device``.

    depth: number of batches in flight (staging buffers are allocated per slot and reused when
    shapes repeat, which they do for every batch but possibly the last one of an epoch).
    """

    def __init__(self, loader: Iterable, device: torch.device, depth: int = 2):
        self.loader = loader
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self._stream: Optional[torch.cuda.Stream] = None
        self._staging: List[dict] = [dict() for _ in range(self.depth)]
        if self.device.type == "cuda":
            self._stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)  # type: ignore[arg-type]

    def _stage(self, slot: int, key: int, t: torch.Tensor) -> torch.Tensor:
        """Pinned copy of a CPU tensor, in a buffer that belongs to (slot, position in batch)."""
        if t.is_cuda or t.is_pinned():
            return t
        buf = self._staging[slot].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._staging[slot][key] = buf
        buf.copy_(t)
        return buf

    def _launch(self, slot: int, batch: Any):
        counter = [0]

        def move(t: torch.Tensor) -> torch.Tensor:
            k = counter[0]
            counter[0] += 1
            return self._stage(slot, k, t).to(self.device, non_blocking=True)

        with torch.cuda.stream(self._stream):
            moved = _map(batch, move)
            done = torch.cuda.Event()
            done.record(self._stream)
        return moved, done

    def __iter__(self) -> Iterator:
        if self._stream is None:
            for batch in self.loader:
                yield _map(batch, lambda t: t.to(self.device))
            return
        it = iter(self.loader)
        queue: List[tuple] = []
        slot = 0
        reuse_guard: List[Optional[torch.cuda.Event]] = [None] * self.depth

        def fill():
            nonlocal slot
            while len(queue) < self.depth:
                try:
                    batch = next(it)
                except StopIteration:
                    return
                if reuse_guard[slot] is not None:
                    reuse_guard[slot].synchronize()  # the copy that last read this slot's pinned buffers is done
                moved, done = self._launch(slot, batch)
                reuse_guard[slot] = done
                queue.append((moved, done))
                slot = (slot + 1) % self.depth

        fill()
        while queue:
            moved, done = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            _map(moved, lambda t: (t.record_stream(cur), t)[1])
            fill()
            yield moved
