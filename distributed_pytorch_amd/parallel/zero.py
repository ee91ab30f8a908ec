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

Under a wrapper built with ``shard_gradients=True`` the split is by RANGE instead of by parameter ("arena mode"):
the wrapper reduce-scatters every gradient bucket in place, this optimizer updates the rank's 1/N range of every
bucket -- fp32 master weights and moments exist for that range only, parameters may straddle range boundaries --
and all-gathers the bucket's parameter buffer, bucket by bucket on the communication stream while the next
bucket is being updated.  Bytes on the links per step: one reduce-scatter plus one all-gather of the model, the
same as the plain wrapper's all-reduce (the ownership mode above costs 1.5x that: all-reduce plus broadcasts).

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
        self.max_grad_norm = max_grad_norm
        self._arena = None
        ddp = self._sharding_wrapper()
        if ddp is not None:
            self._init_arena_mode(ddp, dict(defaults))
            return
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

    # ------------------------------------------------------------------ arena mode (wrapper with shard_gradients=True)
    def _sharding_wrapper(self):
        for p in self._all:
            ref = getattr(p, "_dpa_ddp", None)
            ddp = ref[0]() if ref is not None else None
            if ddp is not None and getattr(ddp, "shard_gradients", False):
                return ddp
        return None

    def _init_arena_mode(self, ddp, defaults: dict) -> None:
        from ..ops import FusedAdamW

        if self.optimizer_class is not FusedAdamW:
            raise ValueError("a wrapper with shard_gradients=True is served by the built-in AdamW update only")
        wrapped = {id(p) for p in ddp._params}
        if {id(p) for p in self._all} != wrapped:
            raise ValueError("arena mode: the optimizer must be given exactly the wrapper's trainable parameters")
        probe = FusedAdamW([torch.zeros(1)], **{k: v for k, v in defaults.items() if k != "master_weights"})
        for g in self.param_groups:
            for k, v in probe.param_groups[0].items():
                if k != "params":
                    g.setdefault(k, v)
        self.world, self.rank = ddp.world_size, ddp._transport.rank()
        self.owner, self._owned_global, self.inner, self._shards = [], [], None, []
        master = bool(defaults.get("master_weights", False))
        index_of = {id(p): i for i, p in enumerate(self._all)}           # optimizer (= state-dict) index of a parameter
        buckets = []
        for b, (gbuf, pbuf, shard, entries) in enumerate(ddp.shard_layout()):
            lo = self.rank * shard
            mine = slice(lo, lo + shard)
            st = {"exp_avg": torch.zeros(shard, dtype=torch.float32, device=gbuf.device),
                  "exp_avg_sq": torch.zeros(shard, dtype=torch.float32, device=gbuf.device)}
            if master and pbuf.dtype != torch.float32:
                st["master"] = pbuf[mine].detach().to(torch.float32).clone()
            buckets.append({"index": b, "grad": gbuf, "param": pbuf, "shard": shard, "lo": lo, "state": st,
                            "entries": [(index_of[id(ddp._params[pi])], off, n) for pi, off, n in entries]})
        self._arena = {"ddp": ddp, "buckets": buckets, "step": 0, "hp": None, "hp_host": None, "master": master, "clip": None}
        ddp.enable_deferred_gradient_wait(True)

    def _arena_hparams(self) -> dict:
        g0 = self.param_groups[0]
        keys = ("lr", "betas", "eps", "weight_decay")
        for g in self.param_groups[1:]:
            if any(g[k] != g0[k] for k in keys):
                raise ValueError("arena mode updates flat ranges that cross parameter boundaries: all parameter groups must "
                                 "share lr / betas / eps / weight_decay (use the wrapper without shard_gradients otherwise)")
        return g0

    @torch.no_grad()
    def _arena_step(self) -> None:
        a = self._arena
        ddp = a["ddp"]
        g0 = self._arena_hparams()
        host = (g0["lr"], g0["betas"][0], g0["betas"][1], g0["eps"], g0["weight_decay"])
        dev = a["buckets"][0]["grad"].device
        a["step"] += 1
        if dev.type != "cuda":
            self._arena_step_cpu(host)
            return
        from .. import _ext

        C = _ext.C()
        if a["hp"] is None:
            vals = torch.tensor(list(host) + [0.0, 0.0, 0.0], dtype=torch.float32)
            vals[5] = torch.tensor(a["step"] - 1, dtype=torch.int32).view(torch.float32)
            a["hp"], a["hp_host"] = vals.to(dev), host
        elif a["hp_host"] != host:
            a["hp"][:5].copy_(torch.tensor(host, dtype=torch.float32), non_blocking=True)
            a["hp_host"] = host
        C.adamw_tick(a["hp"])
        grad_scale = None
        if self.max_grad_norm is not None:
            # the global norm: every rank sums the squares of its own ranges, one small all-reduce adds them up
            ddp.wait_for_gradients()
            if a["clip"] is None:
                z = lambda: torch.zeros(1, dtype=torch.float32, device=dev)
                a["clip"] = {"sumsq": z(), "coef": z(), "norm": z()}
            c = a["clip"]
            c["sumsq"].zero_()
            C.grad_sumsq([bk["grad"][bk["lo"]:bk["lo"] + bk["shard"]] for bk in a["buckets"]], c["sumsq"])
            if self.world > 1:
                pg.all_reduce(c["sumsq"], "sum")
            C.clip_coef(c["sumsq"], float(self.max_grad_norm), c["coef"], c["norm"])
            self._last_norm = c["norm"]
            grad_scale = c["coef"]
        for bk in a["buckets"]:
            ddp.wait_bucket(bk["index"])                       # this bucket's reduce-scatter has landed
            sl = slice(bk["lo"], bk["lo"] + bk["shard"])
            st = bk["state"]
            C.adamw_step([bk["param"][sl]], [bk["grad"][sl]], [st["exp_avg"]], [st["exp_avg_sq"]], [st.get("master")], a["hp"], grad_scale)
            ddp.all_gather_parameters(bk["index"])             # runs on the communication stream under the next bucket's update
        ddp.wait_for_gradients()                               # the forward pass that follows reads every parameter

    def _arena_step_cpu(self, host) -> None:
        """The same update on CPU tensors (the gloo plumbing configuration): torch.optim.AdamW's arithmetic, range-wise."""
        import math

        a = self._arena
        ddp = a["ddp"]
        lr, b1, b2, eps, wd = host
        t = a["step"]
        ddp.wait_for_gradients()
        scale = 1.0
        if self.max_grad_norm is not None:
            sq = torch.zeros(1, dtype=torch.float32)
            for bk in a["buckets"]:
                sq += (bk["grad"][bk["lo"]:bk["lo"] + bk["shard"]].float() ** 2).sum()
            if self.world > 1:
                pg.all_reduce(sq, "sum")
            norm = torch.sqrt(sq)
            self._last_norm = norm
            scale = float(min(1.0, self.max_grad_norm / (float(norm) + 1e-6)))
        for bk in a["buckets"]:
            sl = slice(bk["lo"], bk["lo"] + bk["shard"])
            st = bk["state"]
            p = bk["param"][sl]
            g = bk["grad"][sl].float() * scale
            w = st["master"] if "master" in st else (p if p.dtype == torch.float32 else p.float())
            w.mul_(1 - lr * wd)
            st["exp_avg"].lerp_(g, 1 - b1)
            st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (st["exp_avg_sq"].sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
            w.addcdiv_(st["exp_avg"], denom, value=-lr / (1 - b1 ** t))
            if w is not p:
                p.copy_(w)
            ddp.all_gather_parameters(bk["index"])
        ddp.wait_for_gradients()

    def _arena_gather_state(self, key: str, bk: dict) -> torch.Tensor:
        """The whole bucket's worth of one state tensor (fp32), assembled from every rank's range, on the CPU."""
        st = bk["state"]
        src = st[key] if key in st else bk["param"][bk["lo"]:bk["lo"] + bk["shard"]].float()
        if self.world == 1:
            return src.detach().cpu().clone()
        parts = pg.all_gather(src.detach().contiguous())
        return torch.cat([t.detach().cpu() for t in parts])

    def _arena_consolidated_state(self) -> Dict[str, Any]:
        a = self._arena
        state: Dict[int, Any] = {}
        keys = ["exp_avg", "exp_avg_sq"] + (["master"] if a["master"] else [])
        for bk in a["buckets"]:
            full = {k: self._arena_gather_state(k, bk) for k in keys}
            for oi, off, n in bk["entries"]:
                p = self._all[oi]
                ent = {"step": torch.tensor(float(a["step"]))}
                for k in keys:
                    if k == "master" and p.dtype == torch.float32:
                        continue
                    ent[k] = full[k][off:off + n].view(p.shape).clone()
                state[oi] = ent
        return {"state": dict(sorted(state.items())), "param_groups": self._groups_as_indices()}

    def _arena_load_consolidated(self, sd: Dict[str, Any]) -> None:
        a = self._arena
        groups = sd["param_groups"]
        if [len(g["params"]) for g in groups] != [len(g["params"]) for g in self.param_groups]:
            raise ValueError("loaded state dict has different parameter groups")
        for g_full, g in zip(self.param_groups, groups):
            for k, v in g.items():
                if k != "params":
                    g_full[k] = v
        steps = {int(float(st["step"])) for st in sd["state"].values() if "step" in st}
        if len(steps) > 1:
            raise ValueError("arena mode keeps one step count for all parameters; the loaded state has several")
        a["step"] = steps.pop() if steps else 0
        a["hp"] = None
        for bk in a["buckets"]:
            lo, hi = bk["lo"], bk["lo"] + bk["shard"]
            st = bk["state"]
            for key in ("exp_avg", "exp_avg_sq", "master"):
                if key in st and key != "master":
                    st[key].zero_()
            if "master" in st:
                st["master"].copy_(bk["param"][lo:hi].float())
            for oi, off, n in bk["entries"]:
                ent = sd["state"].get(oi)
                s0, s1 = max(off, lo), min(off + n, hi)      # the part of this parameter inside the rank's range
                if ent is None or s0 >= s1:
                    continue
                for key in ("exp_avg", "exp_avg_sq", "master"):
                    if key in st and key in ent:
                        st[key][s0 - lo:s1 - lo].copy_(ent[key].reshape(-1)[s0 - off:s1 - off].to(torch.float32))

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
        if self._arena is not None:
            if closure is not None:
                with torch.enable_grad():
                    loss = closure()
            self._arena_step()
            return loss
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
        if self._arena is not None:
            return getattr(self, "_last_norm", None)
        return getattr(self.inner, "last_grad_norm", None)

    # ------------------------------------------------------------------ checkpointing
    def state_dict(self) -> Dict[str, Any]:
        """This rank's shard: the inner optimizer's state with the global indices of its parameters."""
        if self._arena is not None:
            a = self._arena
            return {"world": self.world, "rank": self.rank, "arena": True, "step": a["step"],
                    "buckets": [{k: v.detach().clone() for k, v in bk["state"].items()} for bk in a["buckets"]],
                    "param_groups": self._groups_as_indices()}
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

    def consolidated_state_dict(self, rank0_only: bool = False) -> Dict[str, Any]:
        """The state dict an unsharded optimizer over the same parameter groups would return (collective: each owner
        broadcasts its shard, serialised with torch.save).  ``rank0_only``: ranks other than 0 take part in the
        collectives but keep nothing -- 12 bytes of optimizer state per parameter are then assembled in host memory
        once per job instead of once per rank (what a checkpoint written by rank 0 needs)."""
        if self._arena is not None:
            return self._arena_consolidated_state()
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
                    if not rank0_only or self.rank == 0:
                        state.update(torch.load(io.BytesIO(payload.numpy().tobytes()), weights_only=True))
                payload = None
        return {"state": dict(sorted(state.items())), "param_groups": self._groups_as_indices()}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        """Accepts this class's own shard format (same world size and rank) or the unsharded layout of
        ``consolidated_state_dict()`` / a plain optimizer, from which every rank takes what it owns."""
        if self._arena is not None:
            if sd.get("arena"):
                a = self._arena
                if sd["world"] != self.world or sd["rank"] != self.rank or len(sd["buckets"]) != len(a["buckets"]):
                    raise ValueError("shard state belongs to another rank or world size; load the consolidated state dict instead")
                a["step"], a["hp"] = int(sd["step"]), None
                for bk, saved in zip(a["buckets"], sd["buckets"]):
                    for k, v in saved.items():
                        bk["state"][k].copy_(v)
                for g_full, g in zip(self.param_groups, sd["param_groups"]):
                    for k, v in g.items():
                        if k != "params":
                            g_full[k] = v
                return
            self._arena_load_consolidated(sd)
            return
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
