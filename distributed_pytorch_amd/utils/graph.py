"""Whole-step hipGraph capture.

The reference's training iteration (reference min_DDP.py:95-104) is a few dozen tiny kernels
when the model is small: with hidden_dim 4096 a step is launch-bound, not bandwidth-bound.
``GraphedStep`` records forward + loss + backward + optimizer update once into a hipGraph
(through ``torch.cuda.CUDAGraph``; every kernel of this framework launches on the current
stream, allocates only through torch's allocator and never synchronises, so all of them are
capturable) and replays it with one launch per step.  FusedAdamW keeps its step counter and
hyper-parameters in device memory, so replays follow a learning-rate schedule.

Inputs are static tensors: write new data into ``static_inputs`` (``copy_``) between replays.
Single-GPU only; the multi-GPU path overlaps RCCL with backward eagerly instead.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


class GraphedStep:
    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, loss_fn: Callable[[torch.nn.Module], torch.Tensor],
                 warmup: int = 3):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static_loss: Optional[torch.Tensor] = None
        self.replays = 0
        # Warm-up on a side stream: lazy state (optimizer moments, workspaces) is allocated here,
        # outside the capture.
        # (the current stream is drained first, so the side stream's wait below is on a finished event: a stream
        # whose hardware queue shares a command-processor pipe with the compute stream's at another priority
        # must never wait for an unfinished event of it -- csrc/comm/comm_stream.h)
        torch.cuda.current_stream().synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(warmup, 1)):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.static_loss = self._eager_step()

    def _eager_step(self) -> torch.Tensor:
        loss = self.loss_fn(self.model)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def step(self) -> torch.Tensor:
        # The replayed update never passes through optimizer.step(): push changed hyper-parameters (a
        # learning-rate schedule edits param_groups on the host) to the device first, and keep the
        # optimizer's host-side step mirrors in line with the device counters afterwards.
        sync = getattr(self.optimizer, "sync_hparams", None)
        if sync is not None:
            sync()
        self.graph.replay()
        self.replays += 1
        note = getattr(self.optimizer, "note_external_steps", None)
        if note is not None:
            note(1)
        return self.static_loss
