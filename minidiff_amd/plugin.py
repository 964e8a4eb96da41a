"""Reference-side plug-in: ``python your_script.py --backend minidiff_amd.plugin``.

minidiff picks its array library by importing a module and taking the first
class in it that subclasses ``minidiff.backend.Backend``
(reference: minidiff/backend/__init__.py:13-19 flag, :43-77 selection, :80-85
copy of every public class attribute into ``minidiff.backend``). This module is
that class for MI355X: the attributes are the libmdhip-backed functions of
:class:`minidiff_amd.hip_backend.HipBackendTable`, ``tensor_class`` is
:class:`minidiff_amd.ndarray.DeviceArray`. Nothing in minidiff is modified.

Import errors inside a backend module are swallowed by the selector
(minidiff/backend/__init__.py:37-40) and it silently falls back to NumPy, so
call :func:`assert_selected` after ``import minidiff`` when the GPU path is
required.
"""
import minidiff.backend as backend

from .hip_backend import HipBackendTable as _table  # not a Backend subclass: invisible to the selector
from .ndarray import DeviceArray as _DeviceArray


class hip_backend(backend.Backend):
    """MI355X (gfx950) backend: see include/mdhip.h for the C-ABI underneath."""


for _name, _value in vars(_table).items():
    if not _name.startswith("_"):
        setattr(hip_backend, _name, _value)
del _name, _value


def assert_selected():
    import minidiff.backend as b

    if getattr(b, "tensor_class", None) is not _DeviceArray:
        raise RuntimeError("minidiff did not select minidiff_amd.plugin (the selector swallows import errors and "
                           "falls back to NumPy); import minidiff_amd.plugin directly to see the underlying error")
