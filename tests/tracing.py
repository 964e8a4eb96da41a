"""Wrap a backend table so every call is logged in the format of the golden
traces (tests/golden/_trace_backend.py produced those from the real reference)."""
import numpy as np

SKIP = {"tensor_shape", "tensor_size", "tensor_ndim", "tensor_dtype", "tensor_item", "repr", "len", "as_numpy", "array",
        "array_interface", "dtype"}


def describe(x):
    if isinstance(x, np.ndarray):
        return ["arr", str(x.dtype), list(x.shape), bool(x.flags.c_contiguous)]
    if hasattr(x, "is_c_contiguous") and hasattr(x, "_buf"):
        return ["arr", str(x.dtype), list(x.shape), bool(x.is_c_contiguous)]
    if isinstance(x, np.generic):
        return ["npscalar", str(x.dtype)]
    if isinstance(x, (bool, int, float)):
        return ["py", type(x).__name__, x]
    if isinstance(x, (tuple, list)):
        return ["seq", [describe(v) for v in x]]
    if x is None:
        return ["none"]
    return ["obj", type(x).__name__]


def traced_table(table):
    log = []
    ns = {}
    for k in dir(table):
        if k.startswith("__"):
            continue
        f = getattr(table, k)
        if callable(f) and not isinstance(f, type) and k not in SKIP and not k.startswith("_"):
            def make(name, fn):
                def traced(*a, **kw):
                    log.append([name, [describe(v) for v in a], {kk: describe(v) for kk, v in sorted(kw.items())}])
                    return fn(*a, **kw)
                traced.__name__ = name
                return staticmethod(traced)
            ns[k] = make(k, f)
        else:
            ns[k] = f if not callable(f) or isinstance(f, type) else staticmethod(f)
    return type("Traced" + table.__name__, (), ns), log
