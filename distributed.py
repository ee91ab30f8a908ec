"""Drop-in ``distributed`` helper module for single-node data-parallel training on MI355X.

The 18 public functions of the reference helper (reference distributed.py:32-187) with the
same names, signatures and single-process fall-backs, so ``import distributed as dist`` in an
existing training script keeps working.  Behind them:

* ``launch``               own spawner; ranks bound to GPUs through ``HIP_VISIBLE_DEVICES``
* ``init_process_group``   own TCP store rendezvous + native RCCL communicator (gloo on CPU)
* ``prepare_ddp_model``    native bucketed reducer overlapping RCCL all-reduce with backward
* ``data_sampler``         index-exact re-implementation of the rank-strided sampler
* ``all_reduce`` / ``reduce`` / ``gather`` / ``sync_params`` / ``barrier``  RCCL over xGMI

Differences from the reference, all deliberate: peer-to-peer transport is never disabled
(the reference sets NCCL_P2P_DISABLE=1, distributed.py:47); the visibility variable is
``HIP_VISIBLE_DEVICES`` (``CUDA_VISIBLE_DEVICES`` still accepted); ``launch`` can start a
multi-process CPU run (``DIST_WORLD_SIZE`` / ``nprocs=``); ``prepare_ddp_model`` ignores
``device_ids`` for CPU modules instead of raising.
"""
from __future__ import annotations

import socket

import torch
from torch.distributed import ReduceOp  # enum kept so reduce(op=ReduceOp.SUM) call sites work

from distributed_pytorch_amd.parallel import launcher as _launcher
from distributed_pytorch_amd.parallel import process_group as _pg
from distributed_pytorch_amd.parallel.ddp import DistributedDataParallel
from distributed_pytorch_amd.parallel.sampler import ShardedSampler

__all__ = [
    "find_free_port", "launch", "init_process_group", "is_dist_avail_and_initialized", "cleanup", "get_rank",
    "get_device", "is_primary", "get_world_size", "data_sampler", "prepare_ddp_model", "all_reduce", "reduce",
    "gather", "sync_params", "barrier", "wait_for_everyone", "print_primary",
]


# launch
def find_free_port():
    """A TCP port that was free a moment ago (reference distributed.py:32-37).

    ``launch`` does not depend on it -- the launcher binds its store to port 0 itself --
    but it stays available for scripts that export ``MASTER_PORT`` by hand.
    """
    probe = socket.socket()
    try:
        probe.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)  # before bind, where it has an effect
        probe.bind(("127.0.0.1", 0))  # port 0: the kernel picks an unused one
        return probe.getsockname()[1]
    finally:
        probe.close()


def launch(worker_fn, *args, nprocs=None):
    """Run ``worker_fn(rank, world_size, *args)`` once per visible GPU
    (reference distributed.py:40-58); see ``parallel.launcher.launch``."""
    return _launcher.launch(worker_fn, *args, nprocs=nprocs)


# distributed training functions
def init_process_group(rank, world_size, backend=None):
    """Join the job (reference distributed.py:62-66).  ``backend``: ``None`` picks RCCL when
    a GPU is present and gloo otherwise; ``"nccl"`` is an alias of ``"rccl"``."""
    _pg.init_process_group(rank, world_size, backend)


def is_dist_avail_and_initialized():
    return _pg.is_initialized()


def cleanup():
    _pg.destroy_process_group()


def get_rank():
    return _pg.get_rank()


def get_device():
    return _pg.get_device()


def is_primary():
    return get_rank() == 0


def get_world_size():
    return _pg.get_world_size()


# ---- input sharding
def data_sampler(dataset, distributed, shuffle):
    if distributed:
        return ShardedSampler(dataset, shuffle=shuffle)
    return None


# ---- data-parallel wrapper
def prepare_ddp_model(model, device_ids, *args, **kwargs):
    """Wrap for gradient synchronisation when more than one rank runs; otherwise return the
    module unchanged, so single-process checkpoints have no ``module.`` prefix
    (reference distributed.py:112-115)."""
    if get_world_size() > 1:
        model = DistributedDataParallel(model, device_ids, *args, **kwargs)
    return model


# ---- collectives
def all_reduce(tensor, op='sum'):
    """In-place all-reduce with a string op, ``'sum'`` or ``'avg'``; returns the same tensor
    (reference distributed.py:119-133).  ``'avg'`` divides in place, so it needs a float tensor."""
    world_size = get_world_size()
    if world_size == 1:
        return tensor
    if op not in ('sum', 'avg'):
        raise ValueError(f'"{op}" is an invalid reduce operation!')
    _pg.all_reduce(tensor, _pg.OP_SUM)
    if op == 'avg':
        tensor /= world_size
    return tensor


def reduce(tensor, op=ReduceOp.SUM):
    """In-place reduction onto rank 0 (a SUM by default, not a mean); every rank gets its
    tensor object back, only rank 0's holds the result (reference distributed.py:136-144)."""
    if get_world_size() == 1:
        return tensor
    _pg.reduce(tensor, dst=0, op=op)
    return tensor


def gather(data):
    """List of ``world_size`` tensors in rank order; filled on rank 0, zeros elsewhere
    (reference distributed.py:147-160)."""
    if get_world_size() == 1:
        return [data]
    return _pg.gather(data, dst=0)


def sync_params(params):
    """
    Overwrite every tensor of ``params`` with rank 0's copy.
    """
    if is_dist_avail_and_initialized():
        with torch.no_grad():
            for p in params:
                _pg.broadcast(p.data if isinstance(p, torch.nn.Parameter) else p, 0)


def barrier():
    if get_world_size() == 1:
        return
    _pg.barrier()


# alias of barrier() kept for the reference's call sites
def wait_for_everyone():
    barrier()


def print_primary(*args, **kwargs):
    if is_primary():
        print(*args, **kwargs)
