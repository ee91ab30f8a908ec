"""Process-group state and collectives.

Two data planes sit behind one API:

``rccl``  GPU tensors.  The native ``RcclComm`` (``csrc/comm``): a communicator created
          from an ``ncclUniqueId`` exchanged through our TCP store, bound to this rank's
          device, with its own high-priority HIP stream.  ``"nccl"`` is accepted as an
          alias because the reference asks for it by that name (distributed.py:63-64).
``gloo``  CPU tensors.  c10d's gloo group, rendezvoused through our store; this is the
          CPU oracle used by the multi-process tests.

Every query and collective degrades to the single-process answer when no group is
initialised, matching reference distributed.py:69-101,119-177.
"""
from __future__ import annotations

import datetime
import os
import signal
import socket
import sys
import threading
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as tdist

from .. import _ext
from . import heartbeat as hb
from .store import Store, TorchStoreAdapter, create_store_from_env

# ncclRedOp_t values
OP_SUM, OP_PROD, OP_MAX, OP_MIN, OP_AVG = 0, 1, 2, 3, 4
_OP_NAMES = {"sum": OP_SUM, "prod": OP_PROD, "product": OP_PROD, "max": OP_MAX, "min": OP_MIN, "avg": OP_AVG, "mean": OP_AVG}


@dataclass
class _State:
    rank: int
    world_size: int
    backend: str
    store: Optional[Store]
    comm: object = None          # _C.RcclComm for the rccl backend
    device: Optional[torch.device] = None
    owns_c10d: bool = False
    timeout_s: float = 300.0
    gen: int = 0
    heartbeat: Optional[hb.Heartbeat] = None
    monitor_stop: Optional[threading.Event] = None
    monitor_store: Optional[Store] = None
    old_sigterm: object = None
    wakeup_pipe: Optional[tuple] = None
    wakeup_prev_fd: Optional[int] = None
    comm_stream_info: Optional[dict] = None  # how the communication stream was chosen (RcclComm.comm_stream_info)


_state: Optional[_State] = None


def debug_sync() -> bool:
    return os.environ.get("DIST_DEBUG_SYNC", "0") == "1"


def _to_op_code(op) -> int:
    """Accepts our string names, ``torch.distributed.ReduceOp`` members or raw ints."""
    if isinstance(op, str):
        try:
            return _OP_NAMES[op.lower()]
        except KeyError:
            raise ValueError(f'"{op}" is an invalid reduce operation!') from None
    if isinstance(op, int) and not isinstance(op, bool):
        return op
    for name, code in (("SUM", OP_SUM), ("PRODUCT", OP_PROD), ("MAX", OP_MAX), ("MIN", OP_MIN), ("AVG", OP_AVG)):
        if op == getattr(tdist.ReduceOp, name):
            return code
    raise ValueError(f"unsupported reduce op {op!r}")


def _to_c10d_op(code: int):
    return {OP_SUM: tdist.ReduceOp.SUM, OP_PROD: tdist.ReduceOp.PRODUCT, OP_MAX: tdist.ReduceOp.MAX, OP_MIN: tdist.ReduceOp.MIN, OP_AVG: tdist.ReduceOp.AVG}[code]


# ------------------------------------------------------------------------------ queries
def is_initialized() -> bool:
    return _state is not None


def get_rank() -> int:
    return 0 if _state is None else _state.rank


def get_world_size() -> int:
    return 1 if _state is None else _state.world_size


def get_backend() -> Optional[str]:
    return None if _state is None else _state.backend


def get_store() -> Optional[Store]:
    return None if _state is None else _state.store


def get_comm():
    return None if _state is None else _state.comm


def local_device_index(rank: Optional[int] = None) -> int:
    """Device index this rank should use.

    Our launcher rotates ``HIP_VISIBLE_DEVICES`` so the rank's GPU is index 0 and says so
    in ``DIST_LOCAL_DEVICE``; under torchrun it is ``LOCAL_RANK``; otherwise the rank
    itself (reference distributed.py:88-91), wrapped into the visible range.
    """
    v = os.environ.get("DIST_LOCAL_DEVICE")
    if v not in (None, ""):
        return int(v)
    v = os.environ.get("LOCAL_RANK")
    if v not in (None, ""):
        idx = int(v)
    else:
        idx = get_rank() if rank is None else rank
    n = torch.cuda.device_count()
    return idx % n if n > 0 else 0


def get_device() -> torch.device:
    if torch.cuda.is_available():
        if _state is not None and _state.device is not None and _state.device.type == "cuda":
            return _state.device
        return torch.device(f"cuda:{local_device_index()}")
    return torch.device("cpu")


def progress_tick() -> None:
    """Counts one entered collective for the hang monitor (see ``heartbeat.py``)."""
    st = _state
    if st is not None and st.heartbeat is not None:
        st.heartbeat.count += 1


# ------------------------------------------------------------------------------ device map
def check_device_map(entries: Sequence[Tuple[int, str, str]]) -> Optional[str]:
    """``entries`` are (rank, host, PCI bus id).  Returns the error text when two ranks of one
    host are bound to the same GPU (RCCL would refuse or hang in ``ncclCommInitRank``), else None."""
    seen = {}
    for rank, host, bus in entries:
        key = (host, bus.lower())
        if key in seen:
            return (f"ranks {seen[key]} and {rank} are both bound to GPU {bus} on {host}: every rank needs its own "
                    f"device (check HIP_VISIBLE_DEVICES / LOCAL_RANK / DIST_GPU_BINDING)")
        seen[key] = rank
    return None


def format_device_map(entries: Sequence[Tuple[int, str, str]], extra: str = "") -> str:
    rows = ", ".join(f"{r}->{bus}" for r, _h, bus in sorted(entries))
    return f"[rccl] rank->GPU (PCI bus id): {rows}{extra}"


def _verify_devices(C, st: Store, state: "_State", dev: torch.device) -> None:
    """Every rank publishes the PCI bus id of its device; rank 0 checks that they are pairwise
    distinct *before* the communicator is created and prints the map plus the transport-relevant
    environment, so a first multi-GPU run that goes wrong says why."""
    gen, rank, world = state.gen, state.rank, state.world_size
    bus = C.RcclComm.device_bus_id(dev.index if dev.index is not None else 0)
    st.set(f"rccl/dev/{gen}/{rank}", f"{socket.gethostname()}|{bus}")
    verdict_key = f"rccl/devcheck/{gen}"
    if rank == 0:
        entries = []
        for r in range(world):
            host, b = st.get(f"rccl/dev/{gen}/{r}", state.timeout_s).decode().split("|", 1)
            entries.append((r, host, b))
        err = check_device_map(entries)
        st.set(verdict_key, err or "ok")
        if world > 1 and os.environ.get("DIST_QUIET_INIT", "0") != "1":
            env = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_P2P_DISABLE",
                                                  "NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS")}
            shown = " ".join(f"{k}={v}" for k, v in env.items() if v is not None)
            sys.stderr.write(format_device_map(entries, f" | rccl {C.RcclComm.version()} | {shown or 'no transport overrides'}") + "\n")
            if env["NCCL_P2P_DISABLE"] not in (None, "0"):
                sys.stderr.write("[rccl] warning: NCCL_P2P_DISABLE is set -- collectives will bounce through host memory "
                                 "instead of xGMI\n")
            sys.stderr.flush()
    verdict = st.get(verdict_key, state.timeout_s).decode()
    if verdict != "ok":
        raise RuntimeError(f"init_process_group: {verdict}")


class _LibraryStdoutToStderr:
    """While active, file descriptor 1 points at stderr.  RCCL prints a banner (HIP / ROCm version, host name,
    library path) to the C stdout when a communicator is created; a program whose stdout is a machine-readable
    result (the benchmark's one JSON line) must not have it there.  C streams are flushed before the descriptor
    is restored, so nothing buffered inside the library lands on the real stdout later."""

    def __enter__(self):
        import ctypes

        self._libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            sys.stdout.flush()
            self._libc.fflush(None)
        finally:
            os.dup2(self._saved, 1)
            os.close(self._saved)
        return False


_INIT_HINT = ("check the rank->GPU map printed above; HSA_ENABLE_IPC_MODE_LEGACY=%s here (hosts whose driver only supports "
              "dmabuf IPC need it exported as 0, or peer-memory handles fail with 'hipIpcGetMemHandle: invalid argument'); "
              "NCCL_DEBUG=INFO shows the transport RCCL chose")


def guarded_call(fn, timeout_s: float, what: str, on_timeout=None):
    """Runs ``fn()`` (a blocking native call that has released the GIL) under a deadline.  Communicator
    creation is a rendezvous of all ranks inside the library: when it never returns, no rank reaches the
    store barrier or starts its heartbeat, so nothing else would ever end the job.  On expiry a line
    saying what was being waited for is written and the process exits with status 87."""
    import threading

    done = threading.Event()

    def expire():
        if done.is_set():
            return
        sys.stderr.write(f"[rank {os.environ.get('RANK', '?')}] {what} did not return within {timeout_s:.0f} s; "
                         + _INIT_HINT % os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "<unset>") + "\n")
        sys.stderr.flush()
        (on_timeout or (lambda: os._exit(87)))()

    timer = threading.Timer(timeout_s, expire)
    timer.daemon = True
    timer.start()
    try:
        return fn()
    finally:
        done.set()
        timer.cancel()


def comm_stream_wants_high_priority() -> bool:
    """``DIST_COMM_STREAM=normal`` (default) | ``high``.  ``high`` asks for a high-priority communication stream; it is
    only used when the start-up probe finds one whose hardware queue does not share a command-processor pipe with
    the compute stream's (a high-priority queue on that pipe starves the compute queue, see comm_stream.h)."""
    v = os.environ.get("DIST_COMM_STREAM", "normal").lower()
    if v not in ("normal", "high"):
        raise ValueError(f"DIST_COMM_STREAM={v!r}: expected 'normal' or 'high'")
    return v == "high"


# ------------------------------------------------------------------------------ lifecycle
def init_process_group(rank: int, world_size: int, backend: Optional[str] = None, *, timeout_s: Optional[float] = None, store: Optional[Store] = None) -> None:
    global _state
    if _state is not None:
        raise RuntimeError("process group is already initialised")
    if backend is None:
        backend = "rccl" if torch.cuda.is_available() else "gloo"
    backend = {"nccl": "rccl"}.get(backend.lower(), backend.lower())
    if backend not in ("rccl", "gloo"):
        raise ValueError(f"unknown backend '{backend}' (rccl | nccl | gloo)")
    world_size = max(int(world_size), 1)
    rank = int(rank)
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} out of range for world size {world_size}")
    if timeout_s is None:
        timeout_s = float(os.environ.get("DIST_TIMEOUT_S", "300"))

    st = store if store is not None else create_store_from_env(rank, world_size, timeout=timeout_s)
    state = _State(rank=rank, world_size=world_size, backend=backend, store=st, timeout_s=timeout_s)
    # Every rank bumps its own counter, so the n-th init on one store gets generation n on
    # all ranks and never reads keys left behind by an earlier group.
    gen = st.add(f"pg/gen/{rank}", 1)
    state.gen = gen

    if backend == "rccl":
        if not torch.cuda.is_available():
            raise RuntimeError("backend 'rccl' needs a GPU; use backend='gloo' on CPU")
        C = _ext.C()
        dev = torch.device(f"cuda:{local_device_index(rank)}")
        torch.cuda.set_device(dev)  # explicit binding; the reference leaves this to NCCL's inference
        state.device = dev
        _verify_devices(C, st, state, dev)
        key = f"rccl/uid/{gen}"
        if rank == 0:
            st.set(key, C.RcclComm.unique_id())
        uid = st.get(key, timeout_s)
        # Communicator is created eagerly so that iteration 0 does not pay for it.
        try:
            with _LibraryStdoutToStderr():
                state.comm = guarded_call(lambda: C.RcclComm(uid, rank, world_size, dev.index if dev.index is not None else 0,
                                                             comm_stream_wants_high_priority()),
                                          timeout_s, f"ncclCommInitRank (rank {rank} of {world_size})")
        except RuntimeError as exc:
            raise RuntimeError(f"{exc}\n  " + _INIT_HINT % os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "<unset>")) from exc
        state.comm.set_debug_sync(debug_sync())
        if world_size == 1 and os.environ.get("DIST_ONE_RANK_IDENTITY", "0") == "1":
            state.comm.set_one_rank_identity(True)   # measure the reducer machinery without the library's one-rank copy kernel
        # The communication stream is chosen here, once per process and device (csrc/comm/comm_stream.h): a stream of
        # normal priority -- or, with DIST_COMM_STREAM=high, of high priority -- that a bounded probe has seen run
        # concurrently with the compute stream.  Every later communicator of the process reuses it.
        info = state.comm.comm_stream_info()
        state.comm_stream_info = info
        if not info["independent"] and rank == 0 and os.environ.get("DIST_QUIET_INIT", "0") != "1":
            sys.stderr.write(f"[rccl] note: no stream independent of the compute stream was found ({info['text']}); gradient "
                             "all-reduces will not overlap the backward pass (is GPU_MAX_HW_QUEUES very small?)\n")
    else:
        state.device = torch.device("cpu") if not torch.cuda.is_available() else torch.device(f"cuda:{local_device_index(rank)}")
        if not tdist.is_initialized():
            tdist.init_process_group(
                "gloo", store=TorchStoreAdapter(st, prefix=f"c10d/{gen}/"), rank=rank, world_size=world_size,
                timeout=datetime.timedelta(seconds=timeout_s),
            )
            state.owns_c10d = True
    _state = state
    # Everyone has joined before anyone proceeds; also surfaces a dead rank at start-up.
    st.barrier(f"init/{gen}", world_size, timeout_s)
    if world_size > 1:
        _start_hang_detection(state)
    _install_sigterm_abort(state)


def _start_hang_detection(state: "_State") -> None:
    """Heartbeat publisher on every rank; the monitor runs in the launcher parent when it hosts
    the store, otherwise (``torch.distributed.run``, hand-exported MASTER_*) on the hosting rank."""
    st = state.store
    enabled = hb.hang_timeout_s() > 0 or hb.stall_timeout_s() > 0
    if enabled:
        state.heartbeat = hb.Heartbeat(st.host, st.port, state.rank, on_abort=lambda _reason: abort())
    if enabled and st.is_host:
        state.monitor_store = Store(st.host, st.port, is_host=False, timeout=10.0)
        mon = hb.HangMonitor(state.monitor_store, state.world_size)

        def on_hang(v):
            sys.stderr.write(f"[hang monitor] {v.message()}; terminating the job\n")
            sys.stderr.flush()
            # the abort key is posted; this rank hosts the store, so it stays up long enough for the other ranks'
            # heartbeat threads (one poll per interval) to read the key, abort their communicators and leave too
            time.sleep(2.5 * hb.heartbeat_interval_s())
            abort()
            os._exit(hb.EXIT_CODE_ABORTED)

        state.monitor_stop = mon.watch_in_thread(on_hang)


def _install_sigterm_abort(state: "_State") -> None:
    """SIGTERM (launcher teardown, a scheduler's time limit) aborts the communicator first, so kernels of a
    collective that a peer never joined are released before the process goes away.  A Python-level handler alone
    would only run when the main thread returns to the interpreter -- never, for a rank blocked inside a native
    wait on a hung collective -- so the signal is also written to a wake-up descriptor that a small thread reads:
    if the process is still there a second after SIGTERM arrived, that thread aborts the communicator and ends the
    process with the status of a SIGTERM death itself."""
    if state.comm is None or threading.current_thread() is not threading.main_thread():
        return

    def _on_term(signum, frame):
        abort()
        prev = state.old_sigterm
        if callable(prev):
            prev(signum, frame)
        else:
            signal.signal(signal.SIGTERM, signal.SIG_DFL)
            os.kill(os.getpid(), signal.SIGTERM)

    try:
        state.old_sigterm = signal.signal(signal.SIGTERM, _on_term)
    except ValueError:
        state.old_sigterm = None
        return
    try:
        r, w = os.pipe()
        os.set_blocking(w, False)
        state.wakeup_prev_fd = signal.set_wakeup_fd(w, warn_on_full_buffer=False)
        state.wakeup_pipe = (r, w)
    except (ValueError, OSError):
        return
    grace_s = float(os.environ.get("DIST_SIGTERM_GRACE_S", "1.0"))

    def watch():
        while True:
            try:
                data = os.read(r, 64)
            except OSError:
                return
            if not data:
                return
            if signal.SIGTERM in data:
                time.sleep(grace_s)
                # still here: the main thread is inside a native call and the Python handler cannot run.  The abort
                # itself may wait for the device (it drains the communicator's streams), so it gets a bounded time.
                worker = threading.Thread(target=abort, name="dist-sigterm-abort", daemon=True)
                worker.start()
                worker.join(grace_s + 1.0)
                os._exit(128 + int(signal.SIGTERM))

    t = threading.Thread(target=watch, name="dist-sigterm-watch", daemon=True)
    t.start()


def _remove_sigterm_abort(st: "_State") -> None:
    if st.old_sigterm is not None:
        try:
            signal.signal(signal.SIGTERM, st.old_sigterm)
        except ValueError:
            pass
    if st.wakeup_pipe is not None:
        try:
            signal.set_wakeup_fd(st.wakeup_prev_fd if st.wakeup_prev_fd is not None else -1)
        except (ValueError, OSError):
            pass
        r, w = st.wakeup_pipe
        st.wakeup_pipe = None
        for fd in (w, r):   # closing the write end makes the watcher's read return b"" and the thread end
            try:
                os.close(fd)
            except OSError:
                pass


def comm_stream_info() -> Optional[dict]:
    """How this process's communication stream was chosen (priority, probe result), or None without RCCL."""
    st = _state
    return None if st is None else st.comm_stream_info


def abort() -> None:
    """``ncclCommAbort`` on the rank's communicator (no-op on gloo / without a group)."""
    st = _state
    if st is not None and st.comm is not None:
        try:
            st.comm.abort()
        except Exception:
            pass


def destroy_process_group() -> None:
    global _state
    st = _state
    if st is None:
        return
    _state = None
    try:
        if st.monitor_stop is not None:
            st.monitor_stop.set()
        if st.heartbeat is not None:
            st.heartbeat.close()
        _remove_sigterm_abort(st)
        if st.comm is not None:
            st.comm.destroy()
        if st.owns_c10d and tdist.is_initialized():
            tdist.destroy_process_group()
    finally:
        if st.store is not None:
            # The hosting rank keeps the server up until every rank has checked out.
            try:
                key = f"shutdown/{st.gen}"
                st.store.add(key, 1)
                if st.store.is_host:
                    import time

                    deadline = time.time() + 10.0
                    while time.time() < deadline and int(st.store.add(key, 0)) < st.world_size:
                        time.sleep(0.005)
            except Exception:
                pass
            if st.monitor_store is not None:
                st.monitor_store.close()
            st.store.close()


# ------------------------------------------------------------------------------ helpers
def _stage_for_rccl(t: torch.Tensor):
    """RCCL needs device memory; a CPU tensor is round-tripped through the rank's GPU."""
    if t.is_cuda:
        return (t if t.is_contiguous() else t.contiguous()), None
    return t.to(_state.device).contiguous(), t


def _finish(staged: torch.Tensor, original_cpu: Optional[torch.Tensor], user: torch.Tensor) -> None:
    if original_cpu is not None:
        original_cpu.copy_(staged)
    elif staged.data_ptr() != user.data_ptr():
        user.copy_(staged)
    if debug_sync() and staged.is_cuda:
        torch.cuda.synchronize(staged.device)


def _host_copy(t: torch.Tensor) -> torch.Tensor:
    """gloo only ever sees CPU tensors: a GPU tensor is copied to the host here (a synchronous copy on the
    current stream) and the result copied back by the caller.  Given the GPU tensor itself, the group stages it
    on side streams that exchange event waits with the compute stream, and on this runtime such a wait on an
    event the other stream has not reached yet has been seen never to return."""
    return t.cpu() if t.is_cuda else t


# ------------------------------------------------------------------------------ collectives
def all_reduce(t: torch.Tensor, op=OP_SUM) -> torch.Tensor:
    code = _to_op_code(op)
    if _state is None or _state.world_size == 1:
        return t
    progress_tick()
    if _state.backend == "rccl":
        s, cpu = _stage_for_rccl(t)
        _state.comm.all_reduce(s, code)
        _finish(s, cpu, t)
    else:
        h = _host_copy(t)
        if code == OP_AVG:  # gloo has no AVG
            tdist.all_reduce(h, op=tdist.ReduceOp.SUM)
            h /= _state.world_size
        else:
            tdist.all_reduce(h, op=_to_c10d_op(code))
        if h is not t:
            t.copy_(h)
    return t


def reduce(t: torch.Tensor, dst: int = 0, op=OP_SUM) -> torch.Tensor:
    code = _to_op_code(op)
    if _state is None or _state.world_size == 1:
        return t
    progress_tick()
    if _state.backend == "rccl":
        s, cpu = _stage_for_rccl(t)
        _state.comm.reduce(s, int(dst), code)
        _finish(s, cpu, t)
    else:
        h = _host_copy(t)
        tdist.reduce(h, dst=dst, op=_to_c10d_op(code))
        if h is not t and _state.rank == dst:
            t.copy_(h)
    return t


def broadcast(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if _state is None or _state.world_size == 1:
        return t
    progress_tick()
    if _state.backend == "rccl":
        s, cpu = _stage_for_rccl(t)
        _state.comm.broadcast(s, int(src))
        _finish(s, cpu, t)
    else:
        h = _host_copy(t)
        tdist.broadcast(h, src)
        if h is not t and _state.rank != src:
            t.copy_(h)
    return t


def gather(t: torch.Tensor, dst: int = 0) -> List[torch.Tensor]:
    """List of ``world_size`` tensors in rank order; real data on ``dst`` only, zeros elsewhere
    (reference distributed.py:147-160)."""
    if _state is None or _state.world_size == 1:
        return [t]
    w = _state.world_size
    progress_tick()
    if _state.backend == "rccl":
        s, cpu = _stage_for_rccl(t)
        # One contiguous [world, ...] slab filled by a single ncclGather; rows are returned as views.
        out = torch.zeros((w,) + tuple(s.shape), dtype=s.dtype, device=s.device)
        _state.comm.gather(s, out if _state.rank == dst else None, int(dst))
        if debug_sync():
            torch.cuda.synchronize(s.device)
        if cpu is not None:
            out = out.cpu()
        return list(out.unbind(0))
    h = _host_copy(t)
    outs = [torch.zeros_like(h) for _ in range(w)]
    if _state.rank == dst:
        tdist.gather(h, gather_list=outs, dst=dst)
    else:
        tdist.gather(h, dst=dst)
    return [o.to(t.device) for o in outs] if h is not t else outs


def all_gather(t: torch.Tensor) -> List[torch.Tensor]:
    if _state is None or _state.world_size == 1:
        return [t]
    w = _state.world_size
    progress_tick()
    if _state.backend == "rccl":
        s, cpu = _stage_for_rccl(t)
        out = torch.empty((w,) + tuple(s.shape), dtype=s.dtype, device=s.device)
        _state.comm.all_gather(s, out)
        if cpu is not None:
            out = out.cpu()
        return list(out.unbind(0))
    h = _host_copy(t)
    outs = [torch.empty_like(h) for _ in range(w)]
    tdist.all_gather(outs, h)
    return [o.to(t.device) for o in outs] if h is not t else outs


def reduce_scatter(t: torch.Tensor, op=OP_SUM) -> torch.Tensor:
    """Reduces ``t`` (first dimension divisible by the world size) over the ranks and returns this rank's
    slice of the result, ``t.shape[0] // world`` rows: the first half of a ring all-reduce, for callers that
    only need (or only want to update) their own shard.  One ``ncclReduceScatter`` on RCCL; gloo has no such
    collective, so that path reduces the whole tensor and cuts the slice out."""
    code = _to_op_code(op)
    if _state is None or _state.world_size == 1:
        return t
    w, r = _state.world_size, _state.rank
    if t.dim() == 0 or t.shape[0] % w:
        raise ValueError(f"reduce_scatter: first dimension of {tuple(t.shape)} is not divisible by the world size {w}")
    rows = t.shape[0] // w
    progress_tick()
    if _state.backend == "rccl":
        s, cpu = _stage_for_rccl(t)
        out = torch.empty((rows,) + tuple(s.shape[1:]), dtype=s.dtype, device=s.device)
        _state.comm.reduce_scatter(s, out, code)
        if debug_sync():
            torch.cuda.synchronize(s.device)
        return out.cpu() if cpu is not None else out
    full = t.cpu() if t.is_cuda else t.clone()
    if code == OP_AVG:
        tdist.all_reduce(full, op=tdist.ReduceOp.SUM)
        full /= w
    else:
        tdist.all_reduce(full, op=_to_c10d_op(code))
    return full[r * rows:(r + 1) * rows].to(t.device, copy=True)


def barrier() -> None:
    if _state is None or _state.world_size == 1:
        return
    progress_tick()
    if _state.backend == "rccl":
        _state.comm.barrier()
    else:
        tdist.barrier()
