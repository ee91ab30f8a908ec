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
# Accepted and without effect here: they describe where torch's wrapper puts outputs / how it checks itself.
_IGNORED_KWARGS = ("output_device", "dim", "check_reduction", "device_mesh", "skip_all_reduce_unused_params")
# Accepted, but they ask torch's wrapper for behaviour this one does not have: said once, on stderr of rank 0.
_WARNED_KWARGS = {
    "static_graph": "the bucket layout is re-planned once from the observed gradient order in any case; nothing else "
                    "of torch's static-graph mode (activation-checkpoint re-entrancy, delayed all-reduce) exists here",
    "delay_all_reduce_named_params": "every parameter is reduced inside backward; none is delayed",
    "param_to_hook_all_reduce": "every parameter is reduced inside backward; none is delayed",
    "mixed_precision": "use reduce_dtype= (gradient all-reduce dtype) and keep bf16 parameters with fp32 master weights "
                       "in the optimizer (FusedAdamW(master_weights=True)) instead",
}
_warned_once = set()
# Positional order of torch's DistributedDataParallel after (module, device_ids): the reference forwards
# ``*args`` to it untouched (reference distributed.py:112-114), so the same call must bind the same way here.
_TORCH_POSITIONAL = ("output_device", "dim", "broadcast_buffers", "init_sync", "process_group", "bucket_cap_mb",
                     "find_unused_parameters", "check_reduction", "gradient_as_bucket_view", "static_graph")


def _mb_env(name: str, default_mb: float) -> int:
    v = os.environ.get(name)
    return int(float(v if v not in (None, "") else default_mb) * (1 << 20))


def default_bucket_bytes(total_grad_bytes: Optional[int] = None) -> int:
    """Steady-state bucket size: ``DIST_BUCKET_MB`` when set, otherwise a quarter of the model's gradient
    bytes clamped to [32, 256] MiB.  On an 8-GPU xGMI mesh each of the 7 links carries 1/8 of a bucket per
    direct reduce-scatter / all-gather step, so buckets are much larger than the 25 MiB NVLink-era default;
    measured with the one-GPU 8-rank load model (``bench.py --emulate-ranks 8``, 300 GB/s bus bandwidth,
    ``profiles/emulated_allreduce_load.md``): Wide MLP (1.09 GB of gradients) 32 / 64 / 128 / 256 MiB caps ->
    90.8 / 91.8 / 93.2 / 95.0 % of the unwrapped step rate once the optimizer waits bucket by bucket, because the
    tail of the last bucket is hidden under the update of the earlier ones and fewer, larger collectives cost
    less launch and scheduling time; GPT-2 small (248 MB) stays at 62 MiB, 99.6 %."""
    v = os.environ.get("DIST_BUCKET_MB")
    if v not in (None, ""):
        return int(float(v) * (1 << 20))
    if total_grad_bytes is None:
        return 64 << 20
    return int(min(256 << 20, max(32 << 20, total_grad_bytes // 4)))


def default_first_bucket_bytes() -> int:
    """The first bucket is small so the first all-reduce starts early in backward."""
    return _mb_env("DIST_FIRST_BUCKET_MB", 4)


def default_max_collective_bytes(bucket_bytes: int) -> int:
    """Largest single all-reduce: a tensor bigger than the bucket cap (a tied embedding table,
    an 8192x8192 layer) closes its bucket far above the cap and is sent as cap-sized slices."""
    v = os.environ.get("DIST_MAX_COLLECTIVE_MB")
    return int(float(v) * (1 << 20)) if v not in (None, "") else int(bucket_bytes)


class DistributedDataParallel(nn.Module):
    def __init__(self, module: nn.Module, device_ids=None, *args, bucket_cap_mb: Optional[float] = None,
                 first_bucket_mb: Optional[float] = None, broadcast_buffers: bool = True,
                 find_unused_parameters: bool = False, gradient_as_bucket_view: bool = True,
                 reduce_dtype=None, init_sync: bool = True, process_group=None, transport=None,
                 rebuild_buckets: Optional[bool] = None, max_collective_mb: Optional[float] = None,
                 shard_gradients: bool = False, **kwargs):
        super().__init__()
        if len(args) > len(_TORCH_POSITIONAL):
            raise TypeError(f"DistributedDataParallel: at most {len(_TORCH_POSITIONAL) + 2} positional arguments")
        if args:
            given = dict(zip(_TORCH_POSITIONAL, args))
            clash = [k for k in given if k in kwargs]
            if clash:
                raise TypeError(f"DistributedDataParallel: got multiple values for argument(s) {clash}")
            broadcast_buffers = given.pop("broadcast_buffers", broadcast_buffers)
            init_sync = given.pop("init_sync", init_sync)
            process_group = given.pop("process_group", process_group)
            bucket_cap_mb = given.pop("bucket_cap_mb", bucket_cap_mb)
            find_unused_parameters = given.pop("find_unused_parameters", find_unused_parameters)
            gradient_as_bucket_view = given.pop("gradient_as_bucket_view", gradient_as_bucket_view)
            kwargs.update(given)  # output_device, dim, check_reduction, static_graph: handled with the keywords below
        for k in list(kwargs):
            if k in _IGNORED_KWARGS:
                kwargs.pop(k)
            elif k in _WARNED_KWARGS:
                v = kwargs.pop(k)
                if v not in (None, False) and k not in _warned_once:
                    _warned_once.add(k)
                    warnings.warn(f"DistributedDataParallel({k}=...) is accepted for compatibility and has no effect: {_WARNED_KWARGS[k]}",
                                  stacklevel=3)
        if kwargs:
            raise TypeError(f"DistributedDataParallel: unexpected keyword argument(s) {sorted(kwargs)}")
        if process_group is not None and "process_group" not in _warned_once:
            _warned_once.add("process_group")
            warnings.warn("DistributedDataParallel(process_group=...) is ignored: gradients are reduced over the group of "
                          "distributed.init_process_group (all ranks)", stacklevel=2)
        self.module = module
        self.device_ids = device_ids  # accepted for API compatibility; the module's own device is used
        self.broadcast_buffers = bool(broadcast_buffers)
        # Unused parameters are always tolerated; with find_unused_parameters=True the ranks also agree on which
        # parameters were unused EVERYWHERE, and those keep grad = None (torch's behaviour) instead of zeros.
        self.find_unused_parameters = bool(find_unused_parameters)
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
        total_grad_bytes = sum(p.numel() * torch.empty((), dtype=self._bucket_dtype(p)).element_size() for p in self._params)
        self._bucket_bytes = int(bucket_cap_mb * (1 << 20)) if bucket_cap_mb is not None else default_bucket_bytes(total_grad_bytes)
        self._first_bucket_bytes = int(first_bucket_mb * (1 << 20)) if first_bucket_mb is not None else default_first_bucket_bytes()
        self._max_collective_bytes = (int(max_collective_mb * (1 << 20)) if max_collective_mb is not None
                                      else default_max_collective_bytes(self._bucket_bytes))
        self._grad_as_view = bool(gradient_as_bucket_view)
        # Sharded gradients (for parallel/zero.py's ShardedOptimizer): every bucket is reduce-scattered in place, so a
        # rank ends a backward pass with the averaged gradient of its own 1/world range of every bucket; parameters are
        # re-homed into flat buffers with the buckets' layout, which the optimizer updates range-wise and all-gathers.
        self.shard_gradients = bool(shard_gradients)
        if self.shard_gradients:
            if not self._grad_as_view or self.reduce_dtype is not None or self.find_unused_parameters:
                raise ValueError("shard_gradients=True needs gradient_as_bucket_view=True, no reduce_dtype and no find_unused_parameters")
            rebuild_buckets = False   # the layout carries the optimizer's state shards: it is fixed at construction
        if rebuild_buckets is None:
            rebuild_buckets = os.environ.get("DIST_REBUILD_BUCKETS", "1") != "0"
        # Like torch's reducer, the bucket layout is re-planned once, after the first synchronised
        # backward pass, from the order in which gradients really became ready.
        self._rebuild_pending = bool(rebuild_buckets)
        self.buckets_rebuilt = False
        self._carried_syncs = 0      # counters of reducers retired by a rebuild
        self._carried_launches = 0
        self._build_buckets(None)
        if self._grad_as_view:
            # Operators with their own weight-gradient kernels write straight into the parameter's bucket slice
            # (ops/grad_sink.py); the lookup goes through self._reducer, so it survives a bucket re-plan.
            import weakref

            from ..ops import grad_sink

            me = weakref.ref(self)

            def provider(i):
                def take():
                    ddp = me()
                    return ddp._reducer.take_grad_sink(i) if ddp is not None else None
                return take

            for i, p in enumerate(self._params):
                grad_sink.register(p, provider(i))
                p._dpa_ddp = (me, i)   # lets a fused optimizer find the wrapper and the parameter's bucket
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

    def _param_metas(self):
        h = _ext.host()
        dtypes: List[torch.dtype] = []
        metas = []
        for p in self._params:
            bd = self._bucket_dtype(p)
            if bd not in dtypes:
                dtypes.append(bd)
            metas.append(h.ParamMeta(p.numel(), torch.empty((), dtype=bd).element_size(), dtypes.index(bd), True))
        return metas, dtypes

    def _build_buckets(self, ready_order: Optional[List[int]]) -> None:
        """Plans the buckets (registration order reversed, or ``ready_order`` when given), allocates one
        flat arena per bucket dtype and creates the reducer over slices of it."""
        h = _ext.host()
        metas, dtypes = self._param_metas()
        opts = h.PlanOptions(self._first_bucket_bytes, self._bucket_bytes, 128, True)
        plan = h.plan_buckets(metas, opts) if ready_order is None else h.plan_buckets_from_order(metas, list(ready_order), opts)
        self._plan = plan
        if self.shard_gradients:
            # every bucket padded to a multiple of world * 64 elements (equal, 128-byte aligned ranges), buckets back to back
            unit = self.world_size * 64
            planned = list(plan.buckets)
            sizes = [-(-b.numel // unit) * unit for b in planned]
            offs, total = [], {}
            for b, n in zip(planned, sizes):
                offs.append(total.get(b.group_key, 0))
                total[b.group_key] = offs[-1] + n
            self._arenas = {key: torch.zeros(n, dtype=dtypes[key], device=self.device) for key, n in total.items()}
            self._param_arenas = {key: torch.zeros(n, dtype=dtypes[key], device=self.device) for key, n in total.items()}
            bufs = [self._arenas[b.group_key].narrow(0, o, n) for b, o, n in zip(planned, offs, sizes)]
            self._param_buffers = [self._param_arenas[b.group_key].narrow(0, o, n) for b, o, n in zip(planned, offs, sizes)]
        else:
            self._arenas: Dict[int, torch.Tensor] = {
                key: torch.zeros(numel, dtype=dtypes[key], device=self.device) for key, numel in plan.arena_numel
            }
            bufs = [self._arenas[b.group_key].narrow(0, b.arena_offset, b.numel) for b in plan.buckets]
        specs = [self._C.BucketSpec(list(b.param_indices), list(b.offsets), buf) for b, buf in zip(plan.buckets, bufs)]
        self._bucket_buffers = [s.buffer for s in specs]
        self._bucket_params = [list(b.param_indices) for b in plan.buckets]
        self._bucket_offsets = [list(b.offsets) for b in plan.buckets]
        self._reducer = self._C.Reducer(self._params, specs, self._transport, self._grad_as_view, True,
                                        0 if self.shard_gradients else int(self._max_collective_bytes),
                                        self.find_unused_parameters and self.world_size > 1, self.shard_gradients)
        self._reducer.set_deferred_wait(getattr(self, "_defer_wait", False))
        if self.shard_gradients:
            with torch.no_grad():   # parameters move into the flat buffers (same offsets as their gradients)
                for pbuf, idxs, poffs in zip(self._param_buffers, self._bucket_params, self._bucket_offsets):
                    for pi, off in zip(idxs, poffs):
                        p = self._params[pi]
                        if not p.is_contiguous():
                            raise ValueError("shard_gradients=True needs contiguous parameters")
                        v = pbuf.narrow(0, off, p.numel()).view(p.shape)
                        v.copy_(p.detach())
                        p.data = v

    def _agreed_ready_order(self) -> List[int]:
        """Rank 0's observed gradient-ready order, on every rank (the layouts must be identical)."""
        order = [int(i) for i in self._reducer.ready_order()]
        if self.world_size == 1:
            return order
        n = len(self._params)
        t = torch.full((n + 1,), -1, dtype=torch.int64)
        t[0] = len(order)
        t[1:1 + len(order)] = torch.tensor(order, dtype=torch.int64)
        if pg.is_initialized():
            pg.broadcast(t, 0)
        elif tdist.is_initialized():
            tdist.broadcast(t, 0)
        k = int(t[0])
        return [int(i) for i in t[1:1 + k]]

    def rebuild_buckets(self, ready_order: Optional[List[int]] = None) -> bool:
        """Re-plans the buckets so that bucket k holds the k-th group of gradients to become ready, re-homes
        the arenas and moves existing ``.grad`` tensors into their new slices.  Returns False when the layout
        was already the one the order asks for.  Collective on all ranks (the order is rank 0's)."""
        order = self._agreed_ready_order() if ready_order is None else [int(i) for i in ready_order]
        h = _ext.host()
        metas, dtypes = self._param_metas()
        probe = h.plan_buckets_from_order(metas, order, h.PlanOptions(self._first_bucket_bytes, self._bucket_bytes, 128, True))
        if [list(b.param_indices) for b in probe.buckets] == self._bucket_params:
            return False
        old_reducer, old_arenas = self._reducer, self._arenas
        old_reducer.detach()
        self._carried_syncs += int(old_reducer.num_backward_syncs())
        self._carried_launches += int(old_reducer.num_bucket_launches())
        self._build_buckets(order)
        if self._grad_as_view:
            # Gradients that were views of the old arenas move into the new ones.
            loc = {pi: (b, j) for b, idxs in enumerate(self._bucket_params) for j, pi in enumerate(idxs)}
            with torch.no_grad():
                for pi, p in enumerate(self._params):
                    g = p.grad
                    if g is None or g.dtype != self._bucket_dtype(p):
                        continue
                    b, j = loc[pi]
                    off = self._plan.buckets[b].offsets[j]
                    v = self._bucket_buffers[b].narrow(0, off, p.numel()).view(p.shape)
                    v.copy_(g)
                    p.grad = v
        del old_reducer, old_arenas
        self.buckets_rebuilt = True
        return True

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
            if self._rebuild_pending and self.require_backward_grad_sync and self._reducer.num_backward_syncs() >= 1:
                self._rebuild_pending = False
                self.rebuild_buckets()
            if self.require_backward_grad_sync:
                pg.progress_tick()  # one entered "collective" per synchronised backward pass
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

    # ------------------------------------------------------------------ torch-DDP surface beyond forward / no_sync
    def register_comm_hook(self, state, hook) -> None:
        """torch's hook signature is ``hook(state, bucket) -> Future``, built on c10d work objects that do not exist
        on this data plane.  What the stock hooks are used for is available directly: the built-in names
        ``"bf16_compress"`` / ``"fp16_compress"`` (or torch's ``bf16_compress_hook`` / ``fp16_compress_hook``
        functions themselves) switch the gradient all-reduce to that dtype -- buckets are re-allocated in it and
        the pack kernel casts on the way in -- and ``"allreduce"`` / ``allreduce_hook`` restores the parameter
        dtype.  Any other callable raises."""
        name = hook if isinstance(hook, str) else getattr(hook, "__name__", "")
        table = {"bf16_compress": torch.bfloat16, "bf16_compress_hook": torch.bfloat16, "fp16_compress": torch.float16,
                 "fp16_compress_hook": torch.float16, "allreduce": None, "allreduce_hook": None}
        if name not in table:
            raise NotImplementedError(
                f"register_comm_hook: arbitrary communication hooks ({name or hook!r}) are not supported; built in are "
                "'allreduce', 'bf16_compress' and 'fp16_compress' (or the torch functions of those names); for anything else "
                "reduce the gradients yourself under no_sync()")
        if self._reducer.num_backward_syncs() + self._carried_syncs > 0 and self._reducer.waits_pending():
            self._reducer.wait_pending()
        self.reduce_dtype = table[name]
        old = self._reducer
        old.detach()
        self._carried_syncs += int(old.num_backward_syncs())
        self._carried_launches += int(old.num_bucket_launches())
        grads = [p.grad for p in self._params]
        for p in self._params:
            p.grad = None          # views of the arenas that are about to be replaced
        self._build_buckets([i for b in self._bucket_params for i in b] if self.buckets_rebuilt else None)
        for p, g in zip(self._params, grads):
            if g is not None:
                p.grad = g.clone()

    @contextlib.contextmanager
    def join(self, divide_by_initial_world_size: bool = True, enable: bool = True, throw_on_early_termination: bool = False):
        """torch's ``join()`` lets ranks with fewer batches leave early while the others keep reducing.  Here every
        rank must run the same number of synchronised backward passes (the sampler pads shards to equal length, as
        the reference's ``DistributedSampler`` does), so the context only checks that on exit: a rank that ran a
        different number of passes than its peers raises instead of leaving the others hanging in a collective."""
        start = self.num_backward_syncs()
        yield self
        if not enable or self.world_size == 1:
            return
        n = torch.tensor([self.num_backward_syncs() - start], dtype=torch.int64)
        lo, hi = n.clone(), n.clone()
        if pg.is_initialized() and pg.get_backend() == "rccl":
            lo, hi = lo.to(self.device), hi.to(self.device)
        pg.all_reduce(lo, "min")
        pg.all_reduce(hi, "max")
        if int(lo.item()) != int(hi.item()):
            raise RuntimeError(f"join(): ranks ran between {int(lo.item())} and {int(hi.item())} synchronised backward passes; "
                               "uneven inputs are not supported -- give every rank the same number of batches")

    # ------------------------------------------------------------------ deferred gradient waits
    def enable_deferred_gradient_wait(self, on: bool = True) -> None:
        """Leaves the wait for the buckets' all-reduces to the optimizer.  By default ``backward()`` returns with
        the compute stream already ordered behind every all-reduce.  With this on (``FusedAdamW`` /
        ``ShardedOptimizer`` with ``overlap_grad_reduce=True`` switch it on themselves) the optimizer waits
        bucket by bucket, right before it updates a bucket's parameters, so the last all-reduces of a step run
        under the update of the first buckets.  Anything else that reads ``.grad`` between ``backward()`` and
        ``step()`` must call ``wait_for_gradients()`` first."""
        self._defer_wait = bool(on)
        self._reducer.set_deferred_wait(self._defer_wait)

    def wait_for_gradients(self) -> None:
        """The current stream waits for every gradient all-reduce still in flight (no-op when none is)."""
        self._reducer.wait_pending()

    def wait_bucket(self, bucket: int) -> None:
        self._reducer.wait_bucket(int(bucket))

    def bucket_of(self, param_index: int) -> int:
        return int(self._reducer.bucket_of(int(param_index)))

    # ------------------------------------------------------------------ sharded gradients (parallel/zero.py)
    def shard_layout(self):
        """Per bucket: (gradient buffer, parameter buffer, elements per rank, [(parameter index, offset, numel)])."""
        if not self.shard_gradients:
            raise RuntimeError("shard_layout() needs shard_gradients=True")
        out = []
        for gbuf, pbuf, idxs, offs in zip(self._bucket_buffers, self._param_buffers, self._bucket_params, self._bucket_offsets):
            out.append((gbuf, pbuf, gbuf.numel() // self.world_size, [(pi, off, self._params[pi].numel()) for pi, off in zip(idxs, offs)]))
        return out

    def all_gather_parameters(self, bucket: int) -> None:
        """In-place all-gather of bucket ``bucket``'s parameter buffer on the communication stream (every rank
        contributes its own range); ordered behind what the current stream has queued so far."""
        self._reducer.all_gather_bucket(int(bucket), self._param_buffers[int(bucket)])

    # ------------------------------------------------------------------ introspection
    @property
    def reducer(self):
        return self._reducer

    def num_backward_syncs(self) -> int:
        """Synchronised backward passes since construction (across bucket rebuilds)."""
        return self._carried_syncs + int(self._reducer.num_backward_syncs())

    def num_bucket_launches(self) -> int:
        return self._carried_launches + int(self._reducer.num_bucket_launches())

    def bucket_sizes(self) -> List[int]:
        return [int(b.numel() * b.element_size()) for b in self._bucket_buffers]

    def grad_arenas(self) -> List[torch.Tensor]:
        return list(self._arenas.values())

    def describe_buckets(self) -> str:
        return repr(self._plan)

    def parameter_checksum(self) -> torch.Tensor:
        """Sum of all parameters in fp64 (a cheap drift indicator; equal sums do not prove equal parameters)."""
        with torch.no_grad():
            return torch.stack([p.detach().double().sum() for p in self._params]).sum()

    def parameter_hashes(self) -> torch.Tensor:
        """One 64-bit value per parameter computed from its exact bit pattern: the raw bytes, read as integers,
        are weighted by position-dependent odd multipliers and summed with wrap-around, so a changed bit, two
        swapped elements or two errors that cancel in a plain sum all change the value."""
        out = []
        with torch.no_grad():
            for p in self._params:
                t = p.detach().contiguous().view(-1)
                if t.element_size() == 2:
                    bits = t.view(torch.int16).to(torch.int64) & 0xFFFF
                elif t.element_size() == 4:
                    bits = t.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
                else:
                    bits = t.view(torch.uint8).to(torch.int64)
                k = torch.arange(bits.numel(), device=bits.device, dtype=torch.int64)
                weights = (k * 2654435761 + 0x9E3779B97F4A7C15 % (1 << 62)) | 1
                out.append((bits * weights).sum() + bits.numel())
        return torch.stack(out)

    def check_replicas_in_sync(self, rtol: float = 0.0) -> bool:
        """True when every rank holds bit-identical parameters: the per-parameter bit-pattern hashes
        (``parameter_hashes``) are compared across ranks with a min / max all-reduce.  ``rtol`` > 0 falls back to
        the tolerance check on the fp64 checksum (replicas that are allowed to differ in the last bits)."""
        if rtol > 0.0:
            c = self.parameter_checksum().to(torch.float64)
            lo, hi = c.clone(), c.clone()
            pg.all_reduce(lo, "min")
            pg.all_reduce(hi, "max")
            return bool((hi - lo).abs() <= rtol * hi.abs())
        h = self.parameter_hashes()
        lo, hi = h.clone(), h.clone()
        pg.all_reduce(lo, "min")
        pg.all_reduce(hi, "max")
        return bool(torch.equal(lo.cpu(), hi.cpu()))

    def diverged_parameters(self) -> List[str]:
        """Names of the parameters whose bit patterns differ between ranks (collective)."""
        h = self.parameter_hashes()
        lo, hi = h.clone(), h.clone()
        pg.all_reduce(lo, "min")
        pg.all_reduce(hi, "max")
        bad = (lo != hi).cpu().tolist()
        return [n for n, b in zip(self._param_names, bad) if b]
