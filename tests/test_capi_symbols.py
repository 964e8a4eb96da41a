"""The C-ABI contract: every function include/mdhip.h declares is exported by the
product library (dlopen only — no compute call, no GPU needed), by the CPU test
double, and is bound by the ctypes layer; and the product library refuses to
initialise without a gfx950 device instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from minidiff_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mdhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdhip_[a-z0-9_]+)\s*\(", text)) - {"mdhip_array", "mdhip_index_plan", "mdhip_vm_program"})


def test_header_and_ctypes_table_agree():
    assert declared_symbols() == _capi.EXPORTED_SYMBOLS


def test_product_library_exports_every_declared_symbol():
    if not os.path.exists(_capi.PRODUCT_LIB):
        pytest.fail("minidiff_amd/libmdhip.so is not built (run __graft_entry__.build())")
    lib = C.CDLL(_capi.PRODUCT_LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.mdhip_target.restype = C.c_char_p
    assert lib.mdhip_target() == b"hip:gfx950"


def test_test_double_exports_every_declared_symbol(lib):
    from conftest import HOST_DOUBLE
    dbl = C.CDLL(HOST_DOUBLE)
    for name in declared_symbols():
        assert hasattr(dbl, name), name
    dbl.mdhip_target.restype = C.c_char_p
    assert dbl.mdhip_target() == b"host"


def test_product_path_has_no_cpu_fallback(on_gpu, monkeypatch):
    """Without the HIP library the loader raises; with it but without a device, init raises."""
    monkeypatch.setattr(_capi, "PRODUCT_LIB", os.path.join(ROOT, "minidiff_amd", "does_not_exist.so"))
    monkeypatch.setattr(_capi, "_LIB", None)
    with pytest.raises(ImportError):
        _capi.load()
    if not on_gpu and os.path.exists(os.path.join(ROOT, "minidiff_amd", "libmdhip.so")):
        plib = _capi.Library(os.path.join(ROOT, "minidiff_amd", "libmdhip.so"))
        with pytest.raises(RuntimeError):
            plib.init(0)
