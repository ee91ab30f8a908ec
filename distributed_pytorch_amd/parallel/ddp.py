"""Data-parallel module wrapper over the native reducer.

What ``torch.nn.parallel.DistributedDataParallel`` is to the reference
(reference distributed.py:112-115), with the same observable contract -- an ``nn.Module``
whose only child is ``module`` (so ``state_dict()`` keys carry the ``module.`` prefix and
torch-DDP checkpoints load both ways), ``no_sync()``, parameters broadcast from rank 0 at
construction, gradients averaged over ranks after ``backward()`` -- but a different engine:
the C++ ``Reducer`` (``csrc/reducer``) packs complete buckets with one HIP launch and
all-reduces them over RCCL on a separate stream while backward continues; ``.grad`` tensors
are views into one flat arena per dtype, and module buffers are views into a flat buffer
that is broadcast in a single collective.
"""
from __future__ import annotations

import contextlib
import os
import warnings
from typing import Dict, List, Optional

import torch
import torch.distributed as tdist
import torch.nn as nn

from .. import _ext
from . import process_group as pg

_REDUCE_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
                  "fp16": torch.float16, "float16": torch.float16}
_IGNORED_KWARGS = ("output_device", "dim", "check_reduction", "static_graph", "delay_all_reduce_named_params",
                   "param_to_hook_all_reduce", "mixed_precision", "device_mesh", "skip_all_reduce_unused_params")


def _mb_env(name: str, default_mb: float) -> int:
    v = os.environ.get(name)
    return int(float(v if v not in (None, "") else default_mb) * (1 << 20))


def default_bucket_bytes() -> int:
    """Steady-state bucket size.  On an 8-GPU xGMI mesh each of the 7 links carries 1/8 of
    a bucket per direct reduce-scatter / all-gather step, so buckets are larger than the
    25 MiB NVLink-era default: 64 MiB leaves 8 MiB per link per step."""
    return _mb_env("DIST_BUCKET_MB", 64)


def default_first_bucket_bytes() -> int:
    """The first bucket is small so the first all-reduce starts early in backward."""
    return _mb_env("DIST_FIRST_BUCKET_MB", 4)


class DistributedDataParallel(nn.Module):
    def __init__(self, module: nn.Module, device_ids=None, *, bucket_cap_mb: Optional[float] = None,
                 first_bucket_mb: Optional[float] = None, broadcast_buffers: bool = True,
                 find_unused_parameters: bool = False, gradient_as_bucket_view: bool = True,
                 reduce_dtype=None, init_sync: bool = True, process_group=None, transport=None, **kwargs):
        super().__init__()
        for k in list(kwargs):
            if k in _IGNORED_KWARGS:
                kwargs.pop(k)
        if kwargs:
            raise TypeError(f"DistributedDataParallel: unexpected keyword argument(s) {sorted(kwargs)}")
        if process_group is not None:
            warnings.warn("process_group= is ignored: the default group of distributed.init_process_group is used")
        self.module = module
        self.device_ids = device_ids  # accepted for API compatibility; the module's own device is used
        self.broadcast_buffers = bool(broadcast_buffers)
        self.find_unused_parameters = bool(find_unused_parameters)  # unused params are always tolerated
        self.require_backward_grad_sync = True

        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not named:
            raise RuntimeError("DistributedDataParallel is not needed when a module doesn't have any parameter that requires a gradient.")
        self._param_names = [n for n, _ in named]
        self._params: List[torch.Tensor] = [p for _, p in named]
        devices = {p.device for p in self._params}
        if len(devices) != 1:
            raise ValueError(f"DistributedDataParallel: parameters live on several devices {devices}; one device per process expected")
        self.device = devices.pop()

        C = _ext.C()
        self._C = C
        if transport is None:
            if not pg.is_initialized():
                raise RuntimeError("init_process_group() must be called before wrapping a model")
            if pg.get_backend() == "rccl":
                if self.device.type != "cuda":
                    raise ValueError("backend 'rccl' needs the module on the GPU (model.to(distributed.get_device()))")
                transport = C.make_rccl_transport(pg.get_comm())
            else:
                transport = C.make_c10d_transport(tdist.group.WORLD)
        self._transport = transport
        self.world_size = transport.world()

        if isinstance(reduce_dtype, str):
            reduce_dtype = _REDUCE_DTYPES[reduce_dtype.lower()]
        if reduce_dtype is None and os.environ.get("DIST_REDUCE_DTYPE"):
            reduce_dtype = _REDUCE_DTYPES[os.environ["DIST_REDUCE_DTYPE"].lower()]
        self.reduce_dtype = reduce_dtype

        self._verify_replicas()
        self._build_buckets(
            int(bucket_cap_mb * (1 << 20)) if bucket_cap_mb is not None else default_bucket_bytes(),
            int(first_bucket_mb * (1 << 20)) if first_bucket_mb is not None else default_first_bucket_bytes(),
            gradient_as_bucket_view,
        )
        self._flatten_buffers()
        if init_sync and self.world_size > 1:
            self._reducer.broadcast_parameters(0)
            self._sync_buffers()

    # ------------------------------------------------------------------ construction helpers
    def _verify_replicas(self) -> None:
        """Every rank must hold the same parameter list (count and sizes)."""
        if self.world_size == 1 or not pg.is_initialized():
            return
        sig = torch.tensor([len(self._params)] + [p.numel() for p in self._params], dtype=torch.int64)
        lo, hi = sig.clone(), sig.clone()
        if pg.get_backend() == "rccl":
            lo, hi = lo.to(self.device), hi.to(self.device)
        pg.all_reduce(lo, "min")
        pg.all_reduce(hi, "max")
        if not torch.equal(lo.cpu(), hi.cpu()):
            raise RuntimeError("DistributedDataParallel: parameter count / sizes differ across ranks")

    def _bucket_dtype(self, p: torch.Tensor) -> torch.dtype:
        if self.reduce_dtype is not None and p.dtype.is_floating_point:
            return self.reduce_dtype
        return p.dtype

    def _build_buckets(self, bucket_bytes: int, first_bucket_bytes: int, grad_as_view: bool) -> None:
        h = _ext.host()
        dtypes: List[torch.dtype] = []
        metas = []
        for p in self._params:
            bd = self._bucket_dtype(p)
            if bd not in dtypes:
                dtypes.append(bd)
            metas.append(h.ParamMeta(p.numel(), torch.empty((), dtype=bd).element_size(), dtypes.index(bd), True))
        plan = h.plan_buckets(metas, h.PlanOptions(first_bucket_bytes, bucket_bytes, 128, True))
        self._plan = plan
        self._arenas: Dict[int, torch.Tensor] = {
            key: torch.zeros(numel, dtype=dtypes[key], device=self.device) for key, numel in plan.arena_numel
        }
        specs = []
        for b in plan.buckets:
            buf = self._arenas[b.group_key].narrow(0, b.arena_offset, b.numel)
            specs.append(self._C.BucketSpec(list(b.param_indices), list(b.offsets), buf))
        self._bucket_buffers = [s.buffer for s in specs]
        self._reducer = self._C.Reducer(self._params, specs, self._transport, bool(grad_as_view), True)

    def _flatten_buffers(self) -> None:
        """Re-homes module buffers into one flat tensor per dtype so that keeping them in
        sync with rank 0 is a single broadcast instead of one per buffer."""
        self._buffer_arenas: List[torch.Tensor] = []
        if self.world_size == 1:
            return
        by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
        for b in self.module.buffers():
            if b.device == self.device and b.numel() > 0:
                by_dtype.setdefault(b.dtype, []).append(b)
        for dt, bufs in by_dtype.items():
            offs, total = [], 0
            align = max(1, 16 // bufs[0].element_size())
            for b in bufs:
                offs.append(total)
                total += (b.numel() + align - 1) // align * align
            flat = torch.zeros(total, dtype=dt, device=self.device)
            for b, o in zip(bufs, offs):
                v = flat.narrow(0, o, b.numel()).view(b.shape)
                v.copy_(b)
                b.data = v
            self._buffer_arenas.append(flat)

    def _sync_buffers(self) -> None:
        for flat in self._buffer_arenas:
            if flat.dtype == torch.bool:
                continue
            pg.broadcast(flat, 0)

    # ------------------------------------------------------------------ nn.Module surface
    def forward(self, *inputs, **kwargs):
        if torch.is_grad_enabled():
            self._reducer.prepare_for_backward(self.require_backward_grad_sync)
            if self.broadcast_buffers and self._buffer_arenas and self.require_backward_grad_sync:
                with torch.no_grad():
                    self._sync_buffers()
        return self.module(*inputs, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradients accumulate locally inside the context; the first backward after it
        reduces the accumulated sum."""
        old = self.require_backward_grad_sync
        self.require_backward_grad_sync = False
        try:
            yield
        finally:
            self.require_backward_grad_sync = old

    # ------------------------------------------------------------------ introspection
    @property
    def reducer(self):
        return self._reducer

    def bucket_sizes(self) -> List[int]:
        return [int(b.numel() * b.element_size()) for b in self._bucket_buffers]

    def grad_arenas(self) -> List[torch.Tensor]:
        return list(self._arenas.values())

    def describe_buckets(self) -> str:
        return repr(self._plan)

    def parameter_checksum(self) -> torch.Tensor:
        """Sum of all parameters in fp64, for divergence checks across ranks."""
        with torch.no_grad():
            return torch.stack([p.detach().double().sum() for p in self._params]).sum()

    def check_replicas_in_sync(self, rtol: float = 0.0) -> bool:
        """All ranks compare a parameter checksum (the debug aid for stream / bucket races)."""
        c = self.parameter_checksum().to(torch.float64)
        lo, hi = c.clone(), c.clone()
        pg.all_reduce(lo, "min")
        pg.all_reduce(hi, "max")
        return bool((hi - lo).abs() <= rtol * hi.abs())
