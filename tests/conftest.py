"""Test plumbing.

`-m gpu`      : parity tests proper — bind the PRODUCT library (HIP, gfx950) and
                call through the C-ABI; they skip when no GPU is visible.
`-m "not gpu"`: oracle vs golden vectors, host logic, symbol export. The shim's
                host logic is exercised through the CPU test double under
                oracle/ (the only place besides smoke()/bench cpu_baseline that
                may touch oracle/).
One process binds ONE build of the C-ABI: the product library when a gfx950
device is present, the test double otherwise.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (MDHIP_HOST_DOUBLE: another build of the double — the ASan + UBSan one of tests/test_sanitized_double.py)
HOST_DOUBLE = os.environ.get("MDHIP_HOST_DOUBLE") or os.path.join(ROOT, "oracle", "_build", "libmdhip_host.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); runs the HIP library")


def _gpu_visible() -> bool:
    if os.environ.get("MDHIP_FORCE_HOST") == "1":
        return False
    if not os.path.exists("/dev/kfd"):
        return False
    from minidiff_amd import _capi
    if not os.path.exists(_capi.PRODUCT_LIB):
        return False
    try:
        lib = _capi.Library(_capi.PRODUCT_LIB)
        lib.init(int(os.environ.get("MDHIP_DEVICE", "0")))
        return True
    except Exception:
        return False


_STATE = {}


def bound_library():
    """Bind (once) and return (lib, is_gpu)."""
    if "lib" not in _STATE:
        from minidiff_amd import _capi
        if _gpu_visible():
            _STATE["lib"] = _capi.load()
            _STATE["gpu"] = True
        else:
            if not os.path.exists(HOST_DOUBLE):
                subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
            _STATE["lib"] = _capi.use_library(HOST_DOUBLE)
            _STATE["gpu"] = False
    return _STATE["lib"], _STATE["gpu"]


@pytest.fixture(scope="session")
def lib():
    return bound_library()[0]


@pytest.fixture
def mdopt(lib):
    """Set entries of the library's option table (csrc/md_options.h) through the C-ABI test hook mdhip_debug_set_option —
    tile forcing, staging scheme, legacy paths — and put the previous values back when the test ends. The environment is
    NOT the way in: without MDHIP_EXPERIMENTS=1 the library ignores its experiment variables."""
    import ctypes as C
    saved = {}

    def set_option(name, value):
        if name not in saved:
            old = C.c_int64()
            lib.debug_get_option(name.encode(), C.byref(old))
            saved[name] = old.value
        lib.debug_set_option(name.encode(), int(value))

    yield set_option
    for name, old in saved.items():
        lib.debug_set_option(name.encode(), old)


@pytest.fixture(scope="session")
def on_gpu():
    return bound_library()[1]


def pytest_collection_modifyitems(config, items):
    _, gpu = bound_library()
    if gpu:
        return
    skip = pytest.mark.skip(reason="no gfx950 device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def engines(lib):
    """(hip-table engine, numpy-oracle engine)."""
    from minidiff_amd.hip_backend import HipBackendTable
    from minidiff_amd.tape import build_engine
    from oracle.numpy_table import NumpyOracleTable
    return build_engine(HipBackendTable, "dev"), build_engine(NumpyOracleTable, "oracle")
