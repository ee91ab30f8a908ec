"""Process fan-out: one worker process per GPU, bound through ``HIP_VISIBLE_DEVICES``.

Covers what ``torch.multiprocessing.spawn`` does for the reference
(reference distributed.py:40-58) and removes the operational problems its README
documents (orphan workers holding the rendezvous port, README.md:121-125):

* the *parent* hosts the rendezvous store on a kernel-assigned port, so there is no
  window between choosing a port and binding it, and the port dies with the launcher;
* every child is started with its GPU first in ``HIP_VISIBLE_DEVICES`` (before the HIP
  runtime initialises), a CPU-core slice as affinity, and ``PR_SET_PDEATHSIG`` so it
  cannot outlive the launcher;
* the first failing child tears the others down and its traceback is re-raised in
  the parent; SIGINT / SIGTERM on the parent are forwarded.

xGMI peer-to-peer stays enabled: nothing here sets ``NCCL_P2P_DISABLE`` (the reference
forces it on, distributed.py:47, which would push every collective through host memory).
"""
from __future__ import annotations

import contextlib
import ctypes
import multiprocessing as mp
import multiprocessing.connection as mpc
import os
import signal
import sys
import threading
import time
import traceback
from typing import Callable, Dict, List, Optional, Sequence

VISIBILITY_VARS = ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")


class ProcessRaisedException(RuntimeError):
    """A worker raised; carries the rank, pid and formatted remote traceback."""

    def __init__(self, msg: str, rank: int, pid: int):
        super().__init__(msg)
        self.rank = rank
        self.error_index = rank  # name used by torch.multiprocessing's exception
        self.pid = pid


class ProcessExitedException(RuntimeError):
    """A worker died without raising (signal, ``sys.exit(n)``, out-of-memory kill ...)."""

    def __init__(self, msg: str, rank: int, pid: int, exit_code: int, signal_name: Optional[str] = None):
        super().__init__(msg)
        self.rank = rank
        self.error_index = rank
        self.pid = pid
        self.exit_code = exit_code
        self.signal_name = signal_name


class ProcessHungException(RuntimeError):
    """The hang monitor ended the job: ``ranks`` stayed behind their peers (see ``heartbeat.py``)."""

    def __init__(self, msg: str, ranks: List[int], counters: Dict[int, int]):
        super().__init__(msg)
        self.ranks = ranks
        self.rank = ranks[0] if ranks else -1
        self.counters = counters


# --------------------------------------------------------------------------- device list
def visible_device_ids(device_count: int) -> Optional[List[str]]:
    """IDs the user pinned through a visibility variable, or ``None`` if none is set.

    ``ROCR_VISIBLE_DEVICES`` filters below HIP, so under it HIP numbers the survivors
    0..n-1 and those are the IDs to hand to children.
    """
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None and val.strip() != "":
            ids = [s.strip() for s in val.split(",") if s.strip() != ""]
            return ids[:device_count] if device_count else ids
    if os.environ.get("ROCR_VISIBLE_DEVICES", "").strip() != "":
        return [str(i) for i in range(device_count)]
    return None


def child_device_env(ids: Sequence[str], rank: int, mode: str) -> Dict[str, Optional[str]]:
    """Environment edits that bind ``rank`` to ``ids[rank]``.

    ``rotate``   : own GPU first, peers after it -> local device index 0, and every peer
                   stays visible so RCCL can open xGMI peer-to-peer mappings to it.
    ``exclusive``: only the own GPU is visible.
    ``index``    : the environment is left alone, the rank uses device index ``rank``.
    """
    if mode == "index":
        return {"DIST_LOCAL_DEVICE": str(rank)}
    if mode == "exclusive":
        order = [ids[rank]]
    elif mode == "rotate":
        order = list(ids[rank:]) + list(ids[:rank])
    else:
        raise ValueError(f"unknown DIST_GPU_BINDING '{mode}' (rotate | exclusive | index)")
    env: Dict[str, Optional[str]] = {"HIP_VISIBLE_DEVICES": ",".join(order), "DIST_LOCAL_DEVICE": "0"}
    if "ROCR_VISIBLE_DEVICES" not in os.environ:
        # HIP honours both spellings; leave exactly one so the two cannot disagree.
        env["CUDA_VISIBLE_DEVICES"] = None
    return env


def cpu_slices(nprocs: int) -> List[List[int]]:
    """Contiguous, equal CPU-core slices of this process's affinity mask, one per rank."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:  # non-Linux
        return [[] for _ in range(nprocs)]
    per = len(cpus) // nprocs
    if per < 1:
        return [[] for _ in range(nprocs)]
    return [cpus[r * per:(r + 1) * per] for r in range(nprocs)]


def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_numa_nodes(root: str = "/") -> List[int]:
    """NUMA node of every GPU in HIP's enumeration order (the KFD topology's GPU nodes, in node order),
    read from sysfs only -- the launcher parent must not initialise the GPU runtime.  -1 = unknown."""
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    try:
        ids = sorted(int(d) for d in os.listdir(base) if d.isdigit())
    except OSError:
        return []
    out = []
    for i in ids:
        props = {}
        try:
            with open(os.path.join(base, str(i), "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue  # a CPU node
        node = -1
        try:
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            addr = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
            with open(os.path.join(root, "sys/bus/pci/devices", addr, "numa_node")) as f:
                node = int(f.read().strip())
        except (OSError, ValueError):
            node = -1
        out.append(node)
    return out


def numa_cpu_slices(rank_gpus: Sequence[int], gpu_nodes: Sequence[int], node_cpus: Dict[int, List[int]],
                    affinity: Sequence[int]) -> Optional[List[List[int]]]:
    """CPU slice per rank taken from the NUMA node its GPU hangs off: the ranks of one node split that
    node's allowed CPUs evenly, in rank order.  ``rank_gpus[r]`` is the physical GPU index of rank r.
    Returns None when the topology is unknown for any rank (the caller falls back to equal contiguous
    slices), or when a node has fewer allowed CPUs than ranks."""
    allowed = set(affinity)
    by_node: Dict[int, List[int]] = {}
    for r, g in enumerate(rank_gpus):
        if g < 0 or g >= len(gpu_nodes) or gpu_nodes[g] < 0 or gpu_nodes[g] not in node_cpus:
            return None
        by_node.setdefault(gpu_nodes[g], []).append(r)
    out: List[List[int]] = [[] for _ in rank_gpus]
    for node, ranks in by_node.items():
        cpus = [c for c in node_cpus[node] if c in allowed]
        per = len(cpus) // len(ranks)
        if per < 1:
            return None
        for k, r in enumerate(ranks):
            out[r] = cpus[k * per:(k + 1) * per]
    return out


def topology_cpu_slices(rank_gpu_ids: Sequence[str], root: str = "/") -> Optional[List[List[int]]]:
    """``numa_cpu_slices`` fed from sysfs; ``rank_gpu_ids`` are the visibility-variable entries of the ranks'
    GPUs (numeric indices only; UUID-style entries give None)."""
    try:
        gpus = [int(g) for g in rank_gpu_ids]
        affinity = sorted(os.sched_getaffinity(0))
    except (ValueError, AttributeError):
        return None
    nodes = gpu_numa_nodes(root)
    if not nodes:
        return None
    node_cpus: Dict[int, List[int]] = {}
    for n in set(nodes):
        try:
            with open(os.path.join(root, f"sys/devices/system/node/node{n}/cpulist")) as f:
                node_cpus[n] = _parse_cpulist(f.read())
        except OSError:
            pass
    return numa_cpu_slices(gpus, nodes, node_cpus, affinity)


# --------------------------------------------------------------------------- child side
def _set_pdeathsig() -> None:
    try:
        libc = ctypes.CDLL("libc.so.6", use_errno=True)
        PR_SET_PDEATHSIG = 1
        libc.prctl(PR_SET_PDEATHSIG, signal.SIGTERM, 0, 0, 0)
    except Exception:
        pass


def _child_entry(worker_fn: Callable, rank: int, args: tuple, err_conn, cpus: List[int], parent_pid: int) -> None:
    _set_pdeathsig()
    if os.getppid() != parent_pid:  # parent died between fork and prctl
        os._exit(1)
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    # Hang diagnosis: with DIST_HANG_DUMP_S=N every rank prints the Python stacks of all its threads
    # to stderr every N seconds (a stuck collective then shows which rank waits where).
    hang_s = os.environ.get("DIST_HANG_DUMP_S", "")
    if hang_s not in ("", "0"):
        import faulthandler

        faulthandler.dump_traceback_later(float(hang_s), repeat=True, file=sys.stderr)
    try:
        worker_fn(rank, *args)
    except KeyboardInterrupt:
        os._exit(130)
    except SystemExit:
        raise
    except BaseException:
        try:
            err_conn.send(traceback.format_exc())
            err_conn.close()
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(1)


# --------------------------------------------------------------------------- parent side
_env_lock = threading.Lock()


@contextlib.contextmanager
def _patched_environ(edits: Dict[str, Optional[str]]):
    """Children started with the spawn method copy ``os.environ`` at ``start()`` time."""
    with _env_lock:
        saved = {k: os.environ.get(k) for k in edits}
        try:
            for k, v in edits.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            yield
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


class ProcessContext:
    def __init__(self, procs, err_conns, grace_s: float = 5.0):
        self.procs = procs
        self.err_conns = err_conns
        self.grace_s = grace_s

    def pids(self):
        return [p.pid for p in self.procs]

    def terminate_all(self, sig=signal.SIGTERM) -> None:
        for p in self.procs:
            if p.is_alive():
                try:
                    os.kill(p.pid, sig)
                except ProcessLookupError:
                    pass
        deadline = time.time() + self.grace_s
        for p in self.procs:
            p.join(max(0.0, deadline - time.time()))
        for p in self.procs:
            if p.is_alive():
                p.kill()
                p.join(2.0)

    def join(self, timeout: Optional[float] = None) -> bool:
        """Wait for one state change.  True when every worker exited cleanly."""
        alive = {p.sentinel: i for i, p in enumerate(self.procs) if p.exitcode is None}
        if not alive:
            return self._check_all()
        mpc.wait(list(alive.keys()), timeout=timeout)
        return self._check_all()

    def _check_all(self) -> bool:
        failed = None
        for i, p in enumerate(self.procs):
            if p.exitcode is not None and p.exitcode != 0:
                failed = i
                break
        if failed is None:
            return all(p.exitcode == 0 for p in self.procs)
        p = self.procs[failed]
        tb = None
        conn = self.err_conns[failed]
        try:
            if conn.poll(0.2):
                tb = conn.recv()
        except (EOFError, OSError):
            tb = None
        self.terminate_all()
        if tb is not None:
            raise ProcessRaisedException(f"\n\n-- Process {failed} terminated with the following error:\n{tb}", failed, p.pid)
        code = p.exitcode
        if code < 0:
            try:
                name = signal.Signals(-code).name
            except ValueError:
                name = f"signal {-code}"
            raise ProcessExitedException(f"process {failed} terminated with signal {name}", failed, p.pid, code, name)
        raise ProcessExitedException(f"process {failed} terminated with exit code {code}", failed, p.pid, code)


def spawn(worker_fn: Callable, args: tuple = (), nprocs: int = 1, *, env_per_rank: Optional[List[Dict[str, Optional[str]]]] = None, pin_cpus: bool = True, join: bool = True, monitor=None,
          cpu_sets: Optional[List[List[int]]] = None):
    """Start ``nprocs`` fresh interpreters running ``worker_fn(rank, *args)``.

    ``worker_fn`` must be importable from the child (module-level function) and the
    calling script needs an ``if __name__ == "__main__":`` guard, exactly as with the
    spawn start method the reference relies on (reference min_DDP.py:133-139).
    """
    ctx = mp.get_context("spawn")
    # cpu_sets: one CPU list per rank chosen by the caller (NUMA-local to the rank's GPU); else equal contiguous slices
    slices = (cpu_sets if cpu_sets is not None else cpu_slices(nprocs)) if pin_cpus else [[] for _ in range(nprocs)]
    procs, conns = [], []
    parent_pid = os.getpid()
    for rank in range(nprocs):
        edits: Dict[str, Optional[str]] = dict(env_per_rank[rank]) if env_per_rank else {}
        edits.setdefault("RANK", str(rank))
        edits.setdefault("LOCAL_RANK", str(rank))
        edits.setdefault("WORLD_SIZE", str(nprocs))
        edits.setdefault("LOCAL_WORLD_SIZE", str(nprocs))
        if slices[rank] and "OMP_NUM_THREADS" not in os.environ:
            edits.setdefault("OMP_NUM_THREADS", str(len(slices[rank])))
        r_conn, w_conn = ctx.Pipe(duplex=False)
        p = ctx.Process(target=_child_entry, args=(worker_fn, rank, tuple(args), w_conn, slices[rank], parent_pid), daemon=False)
        with _patched_environ(edits):
            p.start()
        w_conn.close()
        procs.append(p)
        conns.append(r_conn)
    pc = ProcessContext(procs, conns)
    if not join:
        return pc

    old_handlers = {}

    forwarding = [False]

    def _forward(signum, _frame):
        if forwarding[0]:  # a second signal while the first is still being handled
            return
        forwarding[0] = True
        pc.terminate_all()
        raise KeyboardInterrupt(f"launcher received signal {signum}")

    in_main = threading.current_thread() is threading.main_thread()
    if in_main:
        for s in (signal.SIGINT, signal.SIGTERM):
            old_handlers[s] = signal.signal(s, _forward)
    try:
        watching = monitor is not None and monitor.enabled
        while not pc.join(timeout=1.0 if watching else None):
            if not watching:
                continue
            verdict = monitor.poll()
            if verdict is not None:
                msg = verdict.message()
                sys.stderr.write(f"[launcher] hang detected: {msg}; terminating all {nprocs} workers\n")
                sys.stderr.flush()
                monitor.post_abort(verdict)  # ranks abort their communicators and exit by themselves ...
                deadline = time.time() + 5.0
                while time.time() < deadline and any(p.is_alive() for p in procs):
                    time.sleep(0.1)
                pc.terminate_all()           # ... whoever is still there is signalled
                raise ProcessHungException(msg, verdict.ranks, verdict.counters)
    except BaseException:
        pc.terminate_all()
        raise
    finally:
        for s, h in old_handlers.items():
            signal.signal(s, h)
    return None


def launch(worker_fn: Callable, *args, nprocs: Optional[int] = None):
    """Run ``worker_fn(rank, world_size, *args)`` once per GPU.

    Same contract as the reference launcher (reference distributed.py:40-58):
    more than one device -> one spawned process per device, the parent blocks until
    all have exited; one device -> in-process call with ``world_size=1``; no device ->
    in-process call with ``world_size=0``.  Additions: ``nprocs=`` / ``DIST_WORLD_SIZE``
    force a process count (this is how a multi-process *CPU* run is started, which the
    reference cannot do), and the GPUs are bound per child (``DIST_GPU_BINDING``).
    """
    import torch

    from .store import Store

    device_count = torch.cuda.device_count()
    override = nprocs if nprocs is not None else os.environ.get("DIST_WORLD_SIZE")
    world_size = int(override) if override not in (None, "") else device_count

    if world_size > 1:
        use_gpu = device_count > 0
        env_per_rank: List[Dict[str, Optional[str]]] = [dict() for _ in range(world_size)]
        if use_gpu:
            ids = visible_device_ids(device_count)
            if ids is None:
                raise ValueError("GPUs not specified. Please set HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is accepted too).")
            mode = os.environ.get("DIST_GPU_BINDING", "rotate")
            if world_size > len(ids):
                # More ranks than GPUs is only meaningful for plumbing tests over gloo (RCCL refuses two
                # ranks on one device), so it has to be asked for.
                if os.environ.get("DIST_ALLOW_GPU_SHARING", "0") != "1":
                    raise ValueError(f"world size {world_size} exceeds the {len(ids)} visible GPU(s): {','.join(ids)} "
                                     "(set DIST_ALLOW_GPU_SHARING=1 to place several gloo ranks on one GPU)")
                for r in range(world_size):
                    env_per_rank[r].update(child_device_env(ids, r % len(ids), mode if mode != "index" else "rotate"))
            else:
                for r in range(world_size):
                    env_per_rank[r].update(child_device_env(ids[:world_size], r, mode))
        store = Store("127.0.0.1", 0, is_host=True, bind_addr="127.0.0.1")
        try:
            # Nothing that changes how the runtime opens peer / IPC mappings is set here: variables such as
            # HSA_ENABLE_IPC_MODE_LEGACY reach the workers exactly as the caller exported them.
            common = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(store.port), "DIST_STORE_HOSTED": "1"}
            for e in env_per_rank:
                e.update(common)
            from .heartbeat import HangMonitor

            cpu_sets = None
            if use_gpu and world_size <= len(ids):
                cpu_sets = topology_cpu_slices(ids[:world_size])  # CPUs next to each rank's GPU, when sysfs tells
            spawn(worker_fn, args=(world_size, *args), nprocs=world_size, env_per_rank=env_per_rank, pin_cpus=os.environ.get("DIST_CPU_AFFINITY", "1") != "0", join=True,
                  monitor=HangMonitor(store, world_size), cpu_sets=cpu_sets)
        finally:
            store.close()
    elif world_size == 1:
        worker_fn(0, world_size, *args)
    else:
        worker_fn(0, world_size, *args)
