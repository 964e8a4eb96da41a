"""GENERATOR-ONLY helper (imported by make_golden.py through the reference's own
`--backend` flag, minidiff/backend/__init__.py:13-19). Wraps the reference's
NumPy alias table so every backend call is logged; no arithmetic of its own."""
import numpy as np

import minidiff.backend as backend
import minidiff.backend.numpy as _ref_numpy_module

LOG = []
ENABLED = [False]
_SKIP = {"tensor_shape", "tensor_size", "tensor_ndim", "tensor_dtype", "tensor_item", "repr", "len", "as_numpy", "array",
         "array_interface", "dtype"}


def describe(x):
    if isinstance(x, np.ndarray):
        return ["arr", str(x.dtype), list(x.shape), bool(x.flags.c_contiguous)]
    if isinstance(x, (np.generic,)):
        return ["npscalar", str(x.dtype)]
    if isinstance(x, (bool, int, float)):
        return ["py", type(x).__name__, x]
    if isinstance(x, (tuple, list)):
        return ["seq", [describe(v) for v in x]]
    if x is None:
        return ["none"]
    return ["obj", type(x).__name__]


def _wrap(name, fn):
    def traced(*a, **k):
        if ENABLED[0]:
            LOG.append([name, [describe(v) for v in a], {kk: describe(v) for kk, v in sorted(k.items())}])
        return fn(*a, **k)
    traced.__name__ = name
    return staticmethod(traced)


_src = vars(_ref_numpy_module.numpy_backend)
_ns = {}
for _k, _v in _src.items():
    if _k.startswith("_"):
        continue
    _f = getattr(_ref_numpy_module.numpy_backend, _k)
    if callable(_f) and not isinstance(_f, type) and _k not in _SKIP:
        _ns[_k] = _wrap(_k, _f)
    else:
        _ns[_k] = _v
del _ref_numpy_module  # keep only ONE Backend subclass visible in this namespace
traced_numpy_backend = type("traced_numpy_backend", (backend.Backend,), _ns)
