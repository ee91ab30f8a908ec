"""Metric logging: JSON lines from the primary rank, device-resident accumulation.

Replaces the reference's per-iteration prints (reference min_DDP.py:110-116, 128-130), each
of which synchronises the device, by a meter that adds on the device and a logger that
touches the host once per report.
"""
from __future__ import annotations

import json
import os
import sys
import time
from typing import Optional

import torch

from ..parallel import process_group as pg


class DeviceMeter:
    """Running sums kept on the device: [loss sum, correct count, sample count, steps]."""

    def __init__(self, device: torch.device):
        self.buf = torch.zeros(4, dtype=torch.float64 if device.type == "cpu" else torch.float32, device=device)

    def update(self, loss: torch.Tensor, correct: Optional[torch.Tensor] = None, n: Optional[int] = None) -> None:
        self.buf[0] += loss.detach().to(self.buf.dtype)
        if correct is not None:
            self.buf[1] += correct.sum()
            self.buf[2] += correct.numel() if n is None else n
        self.buf[3] += 1

    def reduce(self) -> dict:
        """One 4-element collective for all metrics (the reference issues three per iteration)."""
        t = self.buf.clone()
        pg.reduce(t, 0, "sum")
        loss_sum, correct, seen, steps = t.tolist()
        w = pg.get_world_size()
        return {"loss": loss_sum / max(steps, 1), "acc": correct / max(seen, 1), "samples": int(seen), "steps": int(steps / w)}

    def reset(self) -> None:
        self.buf.zero_()


class JsonlLogger:
    def __init__(self, path: Optional[str] = None, stream=None):
        self.path = path
        self.stream = stream if stream is not None else sys.stdout
        if path and pg.get_rank() == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def log(self, **fields) -> None:
        if pg.get_rank() != 0:
            return
        fields.setdefault("time", round(time.time(), 3))
        line = json.dumps(fields)
        if self.path:
            with open(self.path, "a") as f:
                f.write(line + "\n")
        else:
            print(line, file=self.stream, flush=True)
