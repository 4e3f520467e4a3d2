import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_once():
    import __graft_entry__ as ge

    ge.build_oracle()


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU oracle (oracle/liblmc_oracle.so). Test infrastructure only."""
    so = os.path.join(ROOT, "oracle", "liblmc_oracle.so")
    if not os.path.exists(so):
        _build_once()
    from tests import _orc

    return _orc.load(so)


@pytest.fixture(scope="session")
def refdrv():
    """The reference's own headers compiled in place (oracle/_ref/librefdrv.so); skip if not built."""
    so = os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/librefdrv.so not built (needs /root/reference at build time)")
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def pathref_path():
    so = os.path.join(ROOT, "oracle", "_ref", "libpathref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libpathref.so not built (needs /root/reference at build time)")
    return so


GOLDEN = os.path.join(ROOT, "tests", "golden")
