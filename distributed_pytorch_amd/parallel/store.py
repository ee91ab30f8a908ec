"""Rendezvous store front-end over the native TCP store (``csrc/store``).

Takes the place of the c10d ``TCPStore`` that ``init_method="env://"`` creates in the
reference (reference distributed.py:65-66).  One process hosts the server -- the launcher
parent when our own spawner is used, otherwise rank 0 -- and every rank is a client.
"""
from __future__ import annotations

import datetime
import os
from typing import Iterable, Optional

import torch.distributed as tdist

from .. import _ext


class Store:
    """Client handle (plus the server object when this process hosts it)."""

    def __init__(self, host: str, port: int, *, is_host: bool = False, timeout: float = 300.0, bind_addr: str = "0.0.0.0"):
        h = _ext.host()
        self._server = None
        if is_host:
            self._server = h.TcpStoreServer(bind_addr, int(port))
            port = self._server.port
        self.host = host
        self.port = int(port)
        self._client = h.TcpStoreClient(host, self.port, float(timeout))

    # -- key/value API ---------------------------------------------------------------
    def set(self, key: str, value) -> None:
        if isinstance(value, str):
            value = value.encode()
        self._client.set(key, bytes(value))

    def get(self, key: str, timeout: float = -1.0) -> bytes:
        return self._client.get(key, timeout)

    def add(self, key: str, delta: int = 1) -> int:
        return self._client.add(key, int(delta))

    def wait(self, keys: Iterable[str], timeout: float = -1.0) -> None:
        self._client.wait(list(keys), timeout)

    def check(self, keys: Iterable[str]) -> bool:
        return self._client.check(list(keys))

    def delete_key(self, key: str) -> bool:
        return self._client.delete_key(key)

    def compare_set(self, key: str, expected, desired) -> bytes:
        enc = lambda v: v.encode() if isinstance(v, str) else bytes(v)
        return self._client.compare_set(key, enc(expected), enc(desired))

    def append(self, key: str, value) -> None:
        if isinstance(value, str):
            value = value.encode()
        self._client.append(key, bytes(value))

    def num_keys(self) -> int:
        return self._client.num_keys()

    def barrier(self, name: str, world: int, timeout: float = -1.0) -> None:
        self._client.barrier(name, int(world), timeout)

    @property
    def timeout(self) -> float:
        return self._client.timeout

    @timeout.setter
    def timeout(self, t: float) -> None:
        self._client.timeout = float(t)

    @property
    def is_host(self) -> bool:
        return self._server is not None

    def close(self) -> None:
        self._client = None
        if self._server is not None:
            self._server.shutdown()
            self._server = None


class TorchStoreAdapter(tdist.Store):
    """Presents our store through the ``torch.distributed.Store`` interface.

    Only used to bootstrap the gloo process group of the CPU backend, so that even the
    CPU path rendezvouses through the native store rather than c10d's TCPStore.
    """

    def __init__(self, store: Store, prefix: str = "c10d/"):
        super().__init__()
        self._s = store
        self._p = prefix
        self._timeout = datetime.timedelta(seconds=store.timeout if store.timeout > 0 else 300)

    def set(self, key, value):
        self._s.set(self._p + key, value)

    def get(self, key):
        return self._s.get(self._p + key, self._timeout.total_seconds())

    def add(self, key, value):
        return self._s.add(self._p + key, value)

    def wait(self, keys, timeout=None):
        t = self._timeout if timeout is None else timeout
        self._s.wait([self._p + k for k in keys], t.total_seconds())

    def check(self, keys):
        return self._s.check([self._p + k for k in keys])

    def delete_key(self, key):
        return self._s.delete_key(self._p + key)

    def compare_set(self, key, expected, desired):
        return self._s.compare_set(self._p + key, expected, desired)

    def num_keys(self):
        return self._s.num_keys()

    def set_timeout(self, timeout):
        self._timeout = timeout

    @property
    def timeout(self):
        return self._timeout

    def append(self, key, value):
        self._s.append(self._p + key, value)

    def multi_get(self, keys):
        return [self.get(k) for k in keys]

    def multi_set(self, keys, values):
        for k, v in zip(keys, values):
            self.set(k, v)

    def has_extended_api(self):
        return True


def _env_int(name: str, default: Optional[int] = None) -> Optional[int]:
    v = os.environ.get(name)
    return int(v) if v not in (None, "") else default


def create_store_from_env(rank: int, world_size: int, timeout: float = 300.0) -> Store:
    """Build the rank's store handle from ``MASTER_ADDR`` / ``MASTER_PORT``.

    Who hosts the server:
      * ``DIST_STORE_HOSTED=1`` (set by our launcher): the launcher parent hosts; every
        rank is a client.
      * under ``torchrun`` the agent already occupies ``MASTER_PORT`` with its own c10d
        store, so rank 0 hosts ours on a kernel-chosen port and publishes that port
        through the agent's store (the only use made of it).
      * otherwise rank 0 hosts on ``MASTER_PORT``.
    """
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if addr == "localhost":
        addr = "127.0.0.1"
    port = _env_int("MASTER_PORT")
    if port is None:
        raise RuntimeError("MASTER_PORT is not set; use distributed.launch() or export MASTER_ADDR/MASTER_PORT")

    if os.environ.get("DIST_STORE_HOSTED") == "1":
        return Store(addr, port, is_host=False, timeout=timeout)

    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() in ("1", "true"):
        agent = tdist.TCPStore(addr, port, world_size, is_master=False, timeout=datetime.timedelta(seconds=timeout))
        run = os.environ.get("TORCHELASTIC_RUN_ID", "run") + "/" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
        key = f"dpa/{run}/store_port"
        if rank == 0:
            st = Store(addr, 0, is_host=True, timeout=timeout)
            agent.set(key, str(st.port))
            return st
        agent.wait([key])
        return Store(addr, int(agent.get(key).decode()), is_host=False, timeout=timeout)

    return Store(addr, port, is_host=(rank == 0), timeout=timeout)
