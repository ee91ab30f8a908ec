"""Test configuration: the ``gpu`` marker and a one-time check that the native modules exist.

Tiers:
  not gpu  -- run anywhere; multi-process paths use gloo with our own launcher / store.
  gpu      -- need one MI355X; every HIP kernel is compared with a plain PyTorch reference
              computed in fp32 / fp64 on the same inputs.
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def _native_built() -> bool:
    pkg = ROOT / "distributed_pytorch_amd"
    return any(pkg.glob("_host*.so")) and any(pkg.glob("_C*.so"))


@pytest.fixture(scope="session", autouse=True)
def native_modules():
    """Builds the extensions once if the tree was checked out without them."""
    if not _native_built():
        import build

        build.build_all()
    from distributed_pytorch_amd import _ext

    _ext.host()
    _ext.C()
    yield


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
