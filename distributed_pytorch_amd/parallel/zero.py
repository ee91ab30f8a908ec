"""Optimizer state sharded over the data-parallel ranks.

The reference gives every rank a full ``AdamW`` over all parameters (reference min_DDP.py:74): with
fp32 master weights and two moments that is 12 bytes of optimizer state per parameter on every GPU,
and every GPU repeats the same update.  ``ShardedOptimizer`` splits the *parameters* over the ranks by
ownership: a rank builds the inner optimizer (``FusedAdamW`` unless told otherwise) over the parameters
it owns only, updates those, and the owners then publish the new values -- one broadcast per owner out
of a flat buffer the owner's parameters live in as views, so nothing is packed or unpacked.  Gradients
still arrive through the wrapper's bucketed all-reduce, so every rank holds the full averaged gradient
and the global gradient norm for clipping needs no further communication.

Per GPU this stores 1/N of the optimizer state and does 1/N of the update arithmetic; the price is one
pass of the parameters over the links per step (2 bytes per parameter for bf16 models, against the
4 + 4 bytes of moments and 4 of master weights that stay sharded).

``state_dict()`` returns the rank's own shard; ``consolidated_state_dict()`` assembles the layout of an
unsharded optimizer (state keyed by the global parameter index) on every rank, and ``load_state_dict``
accepts that layout on any world size, so a run can be resumed on a different number of GPUs or with
the plain optimizer.
"""
from __future__ import annotations

import io
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import process_group as pg

_ALIGN_BYTES = 256  # every parameter starts on a 256-byte boundary of its flat buffer (vector loads, LDS-DMA rows)


def partition_by_size(sizes: Sequence[int], world: int) -> List[int]:
    """Owner rank of every parameter: largest first onto the rank with the least elements so far (ties to
    the lower rank), which every rank computes identically from the sizes alone."""
    load = [0] * max(world, 1)
    owner = [0] * len(sizes)
    for i in sorted(range(len(sizes)), key=lambda j: (-int(sizes[j]), j)):
        r = min(range(len(load)), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += int(sizes[i])
    return owner


class ShardedOptimizer(torch.optim.Optimizer):
    """An ``Optimizer`` to the outside (``param_groups`` hold ALL parameters, so learning-rate schedulers and
    ``zero_grad`` work as usual); the state lives in the inner optimizer over the owned parameters."""

    def __init__(self, params: Iterable, optimizer_class=None, *, max_grad_norm: Optional[float] = None,
                 flatten: bool = True, **defaults):
        from ..ops import FusedAdamW

        self.optimizer_class = optimizer_class or FusedAdamW
        super().__init__(params, dict(defaults))  # validates the groups (duplicates, empty list) and fills in the given defaults
        self._all: List[torch.Tensor] = [p for g in self.param_groups for p in g["params"]]
        self.world = pg.get_world_size() if pg.is_initialized() else 1
        self.rank = pg.get_rank() if self.world > 1 else 0
        self.owner = partition_by_size([p.numel() for p in self._all], self.world)
        self.max_grad_norm = max_grad_norm

        inner_groups, i = [], 0
        for g in self.param_groups:
            mine = []
            for p in g["params"]:
                if self.owner[i] == self.rank:
                    mine.append(p)
                i += 1
            inner_groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": mine})
        self._owned_global = [i for i, o in enumerate(self.owner) if o == self.rank]
        self.inner = None
        if self._owned_global:
            kw = dict(defaults)
            if max_grad_norm is not None and self._fused_clip():
                kw["max_grad_norm"] = max_grad_norm
            self.inner = self.optimizer_class(inner_groups, **kw)
            if max_grad_norm is not None and self._fused_clip():
                # every rank holds the whole averaged gradient: the norm is taken over all parameters, not the shard
                self.inner.clip_params = self._all
            for g_full, g_in in zip(self.param_groups, self.inner.param_groups):  # defaults filled in by the inner class
                for k, v in g_in.items():
                    if k != "params":
                        g_full.setdefault(k, v)
        self._shards: List[Tuple[int, torch.Tensor, list]] = []
        if self.world > 1:
            self._build_shards(flatten)

    def _fused_clip(self) -> bool:
        from ..ops import FusedAdamW

        return isinstance(self.optimizer_class, type) and issubclass(self.optimizer_class, FusedAdamW)

    # ------------------------------------------------------------------ layout
    @torch.no_grad()
    def _build_shards(self, flatten: bool) -> None:
        for r in range(self.world):
            by_kind: Dict[tuple, List[torch.Tensor]] = {}
            for p, o in zip(self._all, self.owner):
                if o == r:
                    by_kind.setdefault((p.dtype, p.device), []).append(p)
            for (dtype, device), ps in by_kind.items():
                align = max(_ALIGN_BYTES // dtype.itemsize, 1)
                offsets, total = [], 0
                for p in ps:
                    offsets.append(total)
                    total += -(-p.numel() // align) * align
                flat = torch.zeros(total, dtype=dtype, device=device)
                entries = []
                for p, off in zip(ps, offsets):
                    view = flat[off:off + p.numel()]
                    rehome = flatten and p.is_contiguous()
                    if rehome:  # the parameter now lives inside the flat buffer
                        view.copy_(p.detach().reshape(-1))
                        p.data = view.view(p.shape)
                    entries.append((p, view, rehome))
                self._shards.append((r, flat, entries))

    def shard_numel(self) -> List[int]:
        """Elements owned per rank (how even the split is)."""
        load = [0] * self.world
        for p, o in zip(self._all, self.owner):
            load[o] += p.numel()
        return load

    # ------------------------------------------------------------------ step
    def _sync_hparams(self) -> None:
        if self.inner is None:
            return
        for g_full, g_in in zip(self.param_groups, self.inner.param_groups):
            for k, v in g_full.items():
                if k != "params":
                    g_in[k] = v

    @torch.no_grad()
    def _publish(self) -> None:
        for r, flat, entries in self._shards:
            loose = [(p, v) for p, v, rehomed in entries if not rehomed]
            if r == self.rank:
                for p, v in loose:
                    v.copy_(p.reshape(-1))
            pg.broadcast(flat, r)
            if r != self.rank:
                for p, v in loose:
                    p.copy_(v.view(p.shape))

    def step(self, closure=None):
        loss = None
        self._sync_hparams()
        if self.max_grad_norm is not None and not self._fused_clip():
            torch.nn.utils.clip_grad_norm_(self._all, self.max_grad_norm)
        if self.inner is not None:
            loss = self.inner.step(closure)
        elif closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._publish()
        return loss

    @property
    def last_grad_norm(self):
        return getattr(self.inner, "last_grad_norm", None)

    # ------------------------------------------------------------------ checkpointing
    def state_dict(self) -> Dict[str, Any]:
        """This rank's shard: the inner optimizer's state with the global indices of its parameters."""
        return {"world": self.world, "rank": self.rank, "owned": list(self._owned_global),
                "inner": self.inner.state_dict() if self.inner is not None else None}

    def _local_state_by_global_index(self) -> Dict[int, Any]:
        if self.inner is None:
            return {}
        sd = self.inner.state_dict()["state"]
        return {self._owned_global[j]: st for j, st in sd.items()}

    def _groups_as_indices(self) -> List[Dict[str, Any]]:
        out, i = [], 0
        for g in self.param_groups:
            n = len(g["params"])
            out.append({**{k: v for k, v in g.items() if k != "params"}, "params": list(range(i, i + n))})
            i += n
        return out

    def consolidated_state_dict(self) -> Dict[str, Any]:
        """The state dict an unsharded optimizer over the same parameter groups would return, assembled on every
        rank (collective: each owner broadcasts its shard, serialised with torch.save)."""
        state: Dict[int, Any] = {}
        if self.world == 1:
            state = {k: _to_cpu(v, copy=True) for k, v in self._local_state_by_global_index().items()}
        else:
            for r in range(self.world):
                if r == self.rank:
                    buf = io.BytesIO()
                    torch.save({k: _to_cpu(v) for k, v in self._local_state_by_global_index().items()}, buf)
                    payload = torch.frombuffer(bytearray(buf.getvalue()), dtype=torch.uint8).clone()
                    size = torch.tensor([payload.numel()], dtype=torch.int64)
                else:
                    size = torch.zeros(1, dtype=torch.int64)
                pg.broadcast(size, r)
                if r != self.rank:
                    payload = torch.empty(int(size.item()), dtype=torch.uint8)
                if payload.numel():
                    pg.broadcast(payload, r)
                    state.update(torch.load(io.BytesIO(payload.numpy().tobytes()), weights_only=True))
        return {"state": dict(sorted(state.items())), "param_groups": self._groups_as_indices()}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        """Accepts this class's own shard format (same world size and rank) or the unsharded layout of
        ``consolidated_state_dict()`` / a plain optimizer, from which every rank takes what it owns."""
        if "inner" in sd and "owned" in sd:
            if sd["world"] != self.world or sd["rank"] != self.rank or list(sd["owned"]) != self._owned_global:
                raise ValueError("shard state belongs to another rank or world size; load the consolidated state dict instead")
            if self.inner is not None:
                self.inner.load_state_dict(sd["inner"])
                self._pull_hparams()
            return
        groups = sd["param_groups"]
        if [len(g["params"]) for g in groups] != [len(g["params"]) for g in self.param_groups]:
            raise ValueError("loaded state dict has different parameter groups")
        for g_full, g in zip(self.param_groups, groups):
            for k, v in g.items():
                if k != "params":
                    g_full[k] = v
        if self.inner is None:
            return
        local = {j: sd["state"][gi] for j, gi in enumerate(self._owned_global) if gi in sd["state"]}
        inner_groups, j = [], 0
        for g_in, g in zip(self.inner.param_groups, groups):
            n = len(g_in["params"])
            inner_groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": list(range(j, j + n))})
            j += n
        self.inner.load_state_dict({"state": local, "param_groups": inner_groups})

    def _pull_hparams(self) -> None:
        for g_full, g_in in zip(self.param_groups, self.inner.param_groups):
            for k, v in g_in.items():
                if k != "params":
                    g_full[k] = v


def _to_cpu(obj, copy: bool = False):
    if torch.is_tensor(obj):
        t = obj.detach().cpu()
        return t.clone() if copy and t.data_ptr() == obj.data_ptr() else t
    if isinstance(obj, dict):
        return {k: _to_cpu(v, copy) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v, copy) for v in obj)
    return obj
