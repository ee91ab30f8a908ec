"""Fused multi-tensor AdamW.

Same update rule and defaults as ``torch.optim.AdamW`` -- which is what the reference
trains with (``AdamW(model.parameters(), 0.0001)``, reference min_DDP.py:74,104) -- but one
HIP launch updates up to 24 tensors, hyper-parameters and the step counter live in device
memory (so a captured hipGraph keeps following a learning-rate schedule), and 16-bit
parameters can carry fp32 master weights inside the optimizer.  Gradient clipping by global
norm is folded in: the clip coefficient is computed on the device and applied inside the
update kernel, with no host synchronisation.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
from torch.optim import Optimizer

from .. import _ext


def _is_dense(t: torch.Tensor) -> bool:
    """True when the tensor's elements occupy one gap-free, non-overlapping run of storage in
    some dimension order (contiguous, channels-last, permuted views of either)."""
    if t.is_contiguous():
        return True
    expect = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz != 1), key=lambda x: x[1]):
        if stride != expect:
            return False
        expect *= size
    return True


def _flat_dense(t: torch.Tensor) -> torch.Tensor:
    """1-D view over the storage of a dense tensor (any memory format, e.g. channels-last conv
    weights); the update is elementwise, so only a consistent element order matters."""
    if t.is_contiguous():
        return t
    return t.as_strided((t.numel(),), (1,), t.storage_offset())


class FusedAdamW(Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 master_weights: bool = False, max_grad_norm: Optional[float] = None, overlap_grad_reduce: bool = False):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameters: {betas}")
        if weight_decay < 0.0:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.master_weights = bool(master_weights)
        self.max_grad_norm = max_grad_norm
        # Under the data-parallel wrapper: wait for the gradient all-reduces bucket by bucket inside step(), in
        # bucket order, instead of all at once at the end of backward(), so the last all-reduces of a step run
        # under the update of the first buckets (parallel/ddp.py: enable_deferred_gradient_wait).
        self.overlap_grad_reduce = bool(overlap_grad_reduce)
        # Parameters whose gradients define the global norm for clipping; None = this optimizer's own.  A sharded
        # optimizer that updates a subset of the model sets it to all parameters (parallel/zero.py).
        self.clip_params: Optional[List[torch.Tensor]] = None
        # Device-side hyper-parameters + step counter, one float32[8] per (group, cohort).  A cohort is the set
        # of a group's parameters that share a step count; its key is the group's tick count minus that step
        # count, which stays constant for a parameter's whole life: 0 for parameters updated from the first
        # step on, -n for parameters restored at step n, +k for a parameter that got its first gradient after
        # k steps.  Almost always there is exactly one cohort per group.
        self._hparams: Dict[tuple, torch.Tensor] = {}
        self._hparams_host: Dict[tuple, tuple] = {}
        self._ticks: Dict[int, int] = {}                  # group index -> updates applied to the group so far
        self._step_i: Dict[torch.Tensor, int] = {}        # parameter -> its step count as a python int
        self._clip: Optional[Dict[str, torch.Tensor]] = None
        self.last_grad_norm: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ state
    def _init_state(self, p: torch.Tensor) -> dict:
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32)  # host-side mirror, kept for state_dict parity
            st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
            if self.master_weights and p.dtype != torch.float32:
                st["master"] = p.detach().to(torch.float32).clone()
        return st

    @staticmethod
    def _host_hparams(group: dict) -> tuple:
        return (group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"])

    def _device_hparams(self, key: tuple, group: dict, device: torch.device, step_before: int) -> torch.Tensor:
        host = self._host_hparams(group)
        hp = self._hparams.get(key)
        if hp is None or hp.device != device:
            vals = torch.tensor(list(host) + [0.0, 0.0, 0.0], dtype=torch.float32)
            vals[5] = torch.tensor(step_before, dtype=torch.int32).view(torch.float32)
            hp = vals.to(device)
            self._hparams[key] = hp
            self._hparams_host[key] = host
        elif self._hparams_host[key] != host:
            hp[:5].copy_(torch.tensor(host, dtype=torch.float32), non_blocking=True)
            self._hparams_host[key] = host
        return hp

    def sync_hparams(self) -> None:
        """Copies changed ``param_groups`` hyper-parameters (a learning-rate schedule) to the device.  ``step()``
        does this itself; a captured graph that replays the update without calling ``step()`` must call it
        before each replay (``GraphedStep`` does)."""
        for (gi, origin), hp in self._hparams.items():
            host = self._host_hparams(self.param_groups[gi])
            if self._hparams_host[(gi, origin)] != host:
                hp[:5].copy_(torch.tensor(host, dtype=torch.float32), non_blocking=True)
                self._hparams_host[(gi, origin)] = host

    def note_external_steps(self, n: int = 1) -> None:
        """Advances the host-side step mirrors by ``n`` updates that ran on the device without ``step()``
        (graph replays), so that ``state_dict()`` reports the true step counts."""
        for gi, group in enumerate(self.param_groups):
            touched = False
            for p in group["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    st["step"] += n
                    if p in self._step_i:
                        self._step_i[p] += n
                    touched = True
            if touched:
                self._ticks[gi] = self._ticks.get(gi, 0) + n

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()

        gpu_groups = []
        cpu_scale = None
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if any(p.grad.is_sparse for p in ps):
                raise RuntimeError("FusedAdamW does not support sparse gradients")
            if ps[0].is_cuda:
                gpu_groups.append((gi, group, ps))
            else:
                if cpu_scale is None:
                    cpu_scale = self._cpu_clip_scale()
                self._step_cpu(group, ps, cpu_scale)

        if not gpu_groups:
            return loss
        C = _ext.C()
        grad_scale = None
        ddp = self._wrapper_of(gpu_groups[0][2]) if self.overlap_grad_reduce else None
        if ddp is not None and not ddp.reducer.deferred_wait():
            ddp.enable_deferred_gradient_wait(True)   # takes effect from the next backward pass on
        if ddp is not None and (self.max_grad_norm is not None or not ddp.reducer.waits_pending()):
            ddp.wait_for_gradients()                  # the global norm needs every gradient
            ddp = None
        if self.max_grad_norm is not None:
            over = self.clip_params if self.clip_params is not None else [p for _, _, ps in gpu_groups for p in ps]
            grad_scale = self._clip_coefficient(C, [p.grad for p in over if p.grad is not None and p.is_cuda])
        for gi, group, ps in gpu_groups:
            states = [self._init_state(p) for p in ps]
            ticks = self._ticks.get(gi, 0)
            cohorts: Dict[int, list] = {}
            for p, st in zip(ps, states):
                if p not in self._step_i:
                    self._step_i[p] = int(st["step"].item())  # restored or fresh; read once, then mirrored as an int
                cohorts.setdefault(ticks - self._step_i[p], []).append((p, st))
            for origin, all_members in cohorts.items():
                hp = self._device_hparams((gi, origin), group, ps[0].device, ticks - origin)
                C.adamw_tick(hp)
                for bucket, members in self._by_bucket(ddp, all_members):
                    if bucket is not None:
                        ddp.wait_bucket(bucket)
                    self._update(C, members, hp, grad_scale)
            self._ticks[gi] = ticks + 1
        if ddp is not None:
            ddp.wait_for_gradients()   # buckets none of whose parameters were updated here
        return loss

    def _update(self, C, members, hp, grad_scale) -> None:
        pv, gv, mv, vv, wv = [], [], [], [], []
        for p, st in members:
            g = p.grad
            if not _is_dense(p):
                raise RuntimeError("FusedAdamW: parameters must be dense (contiguous in some memory format)")
            if g.stride() != p.stride():
                g = torch.empty_like(p).copy_(g)  # bring the gradient into the parameter's element order
            pv.append(_flat_dense(p))
            gv.append(_flat_dense(g))
            mv.append(_flat_dense(st["exp_avg"]))
            vv.append(_flat_dense(st["exp_avg_sq"]))
            wv.append(_flat_dense(st["master"]) if "master" in st else None)
        C.adamw_step(pv, gv, mv, vv, wv, hp, grad_scale)
        for p, st in members:
            st["step"] += 1
            self._step_i[p] += 1

    @staticmethod
    def _wrapper_of(params):
        for p in params:
            ref = getattr(p, "_dpa_ddp", None)
            if ref is not None:
                return ref[0]()
        return None

    @staticmethod
    def _by_bucket(ddp, members):
        """``members`` split by gradient bucket, in bucket (= all-reduce launch) order; one group without a
        bucket when the deferred waits are not in use."""
        if ddp is None:
            return [(None, members)]
        groups: Dict[int, list] = {}
        for p, st in members:
            ref = getattr(p, "_dpa_ddp", None)
            b = ddp.bucket_of(ref[1]) if ref is not None and ref[0]() is ddp else -1
            groups.setdefault(b, []).append((p, st))
        return [(b if b >= 0 else None, groups[b]) for b in sorted(groups)]

    def _clip_coefficient(self, C, grads: List[torch.Tensor]) -> torch.Tensor:
        dev = grads[0].device
        if self._clip is None or self._clip["sumsq"].device != dev:
            z = lambda: torch.zeros(1, dtype=torch.float32, device=dev)
            self._clip = {"sumsq": z(), "coef": z(), "norm": z()}
        c = self._clip
        c["sumsq"].zero_()
        C.grad_sumsq([_flat_dense(g) if _is_dense(g) else g.contiguous() for g in grads], c["sumsq"])
        C.clip_coef(c["sumsq"], float(self.max_grad_norm), c["coef"], c["norm"])
        self.last_grad_norm = c["norm"]
        return c["coef"]

    def _cpu_clip_scale(self) -> float:
        """Clip coefficient of the CPU path: one global norm over all groups (or ``clip_params``)."""
        if self.max_grad_norm is None:
            return 1.0
        over = self.clip_params if self.clip_params is not None else [p for g in self.param_groups for p in g["params"]]
        norm = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in over if p.grad is not None and not p.is_cuda))
        self.last_grad_norm = norm
        return float(min(1.0, self.max_grad_norm / (norm + 1e-6)))

    def _step_cpu(self, group: dict, ps: List[torch.Tensor], scale: float = 1.0) -> None:
        """Reference arithmetic on CPU tensors (the CPU oracle path): identical formula."""
        lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
        for p in ps:
            st = self._init_state(p)
            st["step"] += 1
            t = int(st["step"].item())
            g = p.grad.float() * scale
            w = st.get("master", p if p.dtype == torch.float32 else p.float())
            w.mul_(1 - lr * wd)
            st["exp_avg"].lerp_(g, 1 - b1)
            st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (st["exp_avg_sq"].sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
            w.addcdiv_(st["exp_avg"], denom, value=-lr / (1 - b1 ** t))
            if w is not p:
                p.copy_(w)

    # ------------------------------------------------------------------ checkpointing
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # Device-side step counters are rebuilt from the restored per-parameter steps.
        self._hparams.clear()
        self._hparams_host.clear()
        self._ticks.clear()
        self._step_i.clear()
        # torch hands tensors that already have the right dtype / device over as they are: cloned here, so that this
        # optimizer never updates the tensors of the dict it was loaded from (a live optimizer's state_dict()).
        for st in self.state.values():
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().to("cpu", torch.float32).clone()
            for k in ("exp_avg", "exp_avg_sq", "master"):
                if k in st:
                    st[k] = st[k].to(torch.float32, copy=True)
