"""Hang detection without touching the GPU: progress heartbeats through the rendezvous store.

The reference inherits c10d's collective timeout from ``init_process_group`` (reference
distributed.py:65-66): a rank that waits too long inside a collective takes the job down.  On
this stack polling GPU events from a watchdog thread is not an option (event and stream queries
proved unreliable next to RCCL), so progress is tracked on the host side only:

* every rank counts the collectives it *enters* (facade collectives, one per reducer-driven
  backward pass) and a CPU-only thread publishes that counter to the store about once a second
  -- the thread makes no HIP call in normal operation;
* a monitor (the launcher parent, which hosts the store; rank 0 under a foreign launcher such as
  ``torch.distributed.run``) reads the counters.  A rank whose counter stays behind the others
  for ``DIST_HANG_TIMEOUT_S`` seconds is the one the job is waiting for: it is named with its
  counter, ``hb/abort`` is posted, and the job is terminated with a non-zero exit status;
* a rank's heartbeat thread that sees ``hb/abort`` aborts its RCCL communicator (which releases
  kernels spinning on a peer that will never arrive) and exits the process; a retry starts
  fresh processes, never a re-exec of one that has used the GPU.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from typing import Callable, Dict, List, Optional, Tuple

ABORT_KEY = "hb/abort"
EXIT_CODE_ABORTED = 86


def hang_timeout_s() -> float:
    """Seconds a rank may stay behind its peers (0 disables).  Default 600, c10d's NCCL default."""
    v = os.environ.get("DIST_HANG_TIMEOUT_S")
    return float(v) if v not in (None, "") else 600.0


def stall_timeout_s() -> float:
    """Seconds ALL ranks may sit on the same counter before the job is declared hung (0 = never,
    the default: equal counters are also what a long phase without collectives looks like)."""
    v = os.environ.get("DIST_STALL_TIMEOUT_S")
    return float(v) if v not in (None, "") else 0.0


def heartbeat_interval_s() -> float:
    v = os.environ.get("DIST_HEARTBEAT_S")
    return float(v) if v not in (None, "") else 1.0


def _key(rank: int) -> str:
    return f"hb/{rank}"


class Heartbeat:
    """Rank side.  ``tick()`` is called by the training thread; a daemon thread publishes."""

    def __init__(self, host: str, port: int, rank: int, *, interval_s: Optional[float] = None,
                 on_abort: Optional[Callable[[str], None]] = None):
        from .store import Store

        self.rank = int(rank)
        self.count = 0
        self.interval_s = heartbeat_interval_s() if interval_s is None else float(interval_s)
        self._on_abort = on_abort
        self._stop = threading.Event()
        self._store = Store(host, port, is_host=False, timeout=10.0)  # own connection: clients are not shared between threads
        self._publish("run")
        self._thread = threading.Thread(target=self._loop, name=f"dpa-heartbeat-{rank}", daemon=True)
        self._thread.start()

    def tick(self) -> None:
        self.count += 1

    def _publish(self, phase: str) -> None:
        self._store.set(_key(self.rank), f"{self.count}|{phase}|{os.getpid()}")

    def _loop(self) -> None:
        while not self._stop.wait(self.interval_s):
            try:
                self._publish("run")
                if self._store.check([ABORT_KEY]):
                    reason = self._store.get(ABORT_KEY, 5.0).decode(errors="replace")
                    self._aborted(reason)
                    return
            except Exception:
                return  # store gone: the job is ending

    def _aborted(self, reason: str) -> None:
        sys.stderr.write(f"[rank {self.rank}] job aborted by the hang monitor: {reason}\n")
        sys.stderr.flush()
        try:
            if self._on_abort is not None:
                self._on_abort(reason)
        finally:
            os._exit(EXIT_CODE_ABORTED)

    def close(self) -> None:
        """Marks the rank as finished (the monitor stops expecting progress from it)."""
        self._stop.set()
        try:
            self._publish("done")
        except Exception:
            pass
        self._thread.join(2.0)
        self._store.close()


class HangVerdict:
    def __init__(self, ranks: List[int], counters: Dict[int, int], waited_s: float, kind: str):
        self.ranks, self.counters, self.waited_s, self.kind = ranks, counters, waited_s, kind

    def message(self) -> str:
        ahead = max(self.counters.values())
        who = ", ".join(f"rank {r} (collective #{self.counters[r]})" for r in self.ranks)
        if self.kind == "lag":
            return (f"no progress for {self.waited_s:.0f} s: {who} never entered collective #{ahead} "
                    f"that the other ranks are waiting in")
        return (f"no rank has entered a collective for {self.waited_s:.0f} s; all at #{ahead}: {who}")


class HangMonitor:
    """Reads the heartbeat counters from the store and decides whether the job hangs."""

    def __init__(self, store, world_size: int, *, timeout_s: Optional[float] = None, stall_s: Optional[float] = None,
                 clock: Callable[[], float] = time.monotonic):
        self.store = store
        self.world = int(world_size)
        self.timeout_s = hang_timeout_s() if timeout_s is None else float(timeout_s)
        self.stall_s = stall_timeout_s() if stall_s is None else float(stall_s)
        self.clock = clock
        self._seen: Dict[int, Tuple[int, float]] = {}  # rank -> (counter, time it last changed)

    @property
    def enabled(self) -> bool:
        return self.timeout_s > 0 or self.stall_s > 0

    def read(self) -> Dict[int, Tuple[int, str]]:
        out = {}
        for r in range(self.world):
            if not self.store.check([_key(r)]):
                continue
            parts = self.store.get(_key(r), 5.0).decode().split("|")
            out[r] = (int(parts[0]), parts[1] if len(parts) > 1 else "run")
        return out

    def observe(self, state: Dict[int, Tuple[int, str]]) -> Optional[HangVerdict]:
        """Pure decision step (unit-testable): feed the current {rank: (counter, phase)}."""
        now = self.clock()
        for r, (c, _) in state.items():
            if r not in self._seen or self._seen[r][0] != c:
                self._seen[r] = (c, now)
        running = {r: c for r, (c, ph) in state.items() if ph != "done"}
        if len(state) < self.world or len(running) < 2:
            return None  # not everybody has joined yet, or (almost) everybody has finished
        lo, hi = min(running.values()), max(running.values())
        behind = [r for r, c in running.items() if c == lo]
        idle = min(now - self._seen[r][1] for r in behind)
        if hi > lo and self.timeout_s > 0 and idle >= self.timeout_s:
            return HangVerdict(sorted(behind), running, idle, "lag")
        if hi == lo and self.stall_s > 0 and idle >= self.stall_s:
            return HangVerdict(sorted(behind), running, idle, "stall")
        return None

    def poll(self) -> Optional[HangVerdict]:
        if not self.enabled:
            return None
        try:
            return self.observe(self.read())
        except Exception:
            return None

    def post_abort(self, verdict: HangVerdict) -> None:
        try:
            self.store.set(ABORT_KEY, verdict.message())
        except Exception:
            pass

    def watch_in_thread(self, on_hang: Callable[[HangVerdict], None], interval_s: float = 1.0) -> threading.Event:
        """For a rank that hosts the store under a foreign launcher: polls until ``stop`` is set."""
        stop = threading.Event()

        def loop():
            while not stop.wait(interval_s):
                v = self.poll()
                if v is not None:
                    self.post_abort(v)
                    on_hang(v)
                    return

        threading.Thread(target=loop, name="dpa-hang-monitor", daemon=True).start()
        return stop
