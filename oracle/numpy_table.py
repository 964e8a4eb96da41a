"""CPU oracle: the reference's NumPy backend, restated as a function table.

TEST INFRASTRUCTURE (see oracle/__init__.py). This is the checker the HIP path is
compared against and the `cpu_baseline` that bench.py times; it is never the
thing shipped or measured as the product.

What it restates: minidiff/backend/numpy.py:14-206 — a class whose attributes
alias NumPy functions one-for-one (the arithmetic of the hot path therefore
lives in NumPy 2.x + OpenBLAS, pinned by the reference at numpy 2.3.1 in
uv.lock:30-31; this container and the GPU box run numpy 2.2.6, both NEP 50).
Pinning: tests/test_oracle_golden.py checks this table, driven by
minidiff_amd.tape, against fixtures generated from the real reference
(tests/golden/make_golden.py) — forward values, gradients and the ordered
backend-call traces.
"""
from __future__ import annotations

import numpy as np

# names that are plain aliases of the same-named NumPy function
_SAME_NAME = """
absolute all any argmax argmin argwhere atleast_1d atleast_2d atleast_3d ceil copy cos cosh exp flip floor invert log
logical_not max mean min prod sign sin sinh squeeze std sum tan tanh transpose add broadcast_to dot equal expand_dims
floor_divide greater greater_equal less less_equal logical_and logical_or logical_xor matmul mod multiply not_equal
power reshape subtract tensordot true_divide clip swapaxes where ones_like ones zeros_like zeros full_like full
concatenate isin unravel_index take_along_axis put_along_axis repeat tile arange stack save load split
""".split()
_RANDOM = "choice rand randint randn binomial permutation shuffle".split()
_DTYPES = "float64 float32 float16 uint64 uint32 uint16 uint8 int64 int32 int16 int8".split()


def _vmap(fun):
    # numpy.py:110-122: rows of the leading axis through apply_along_axis
    def mapped(arr):
        shp = arr.shape
        flat = arr.reshape(shp[0], -1)
        return np.apply_along_axis(lambda row: fun(row.reshape(shp[1:])), 1, flat)
    return mapped


def _array(data, dtype=None, copy=None):
    if dtype != data.dtype:
        if not copy:
            raise ValueError("attempted cast, but copies are not permitted")
        return data.astype(dtype=dtype)
    return data.copy() if copy else data


def _build():
    ns = {n: staticmethod(getattr(np, n)) for n in _SAME_NAME}
    ns.update({n: staticmethod(getattr(np.random, n)) for n in _RANDOM})
    ns.update({n: getattr(np, n) for n in _DTYPES})
    ns.update(
        tensor_constructor=staticmethod(np.array),
        tensor_class=np.ndarray,
        flatten=staticmethod(lambda a, order="C": a.flatten(order=order)),
        ravel=staticmethod(lambda a, order="C": a.ravel(order=order)),
        astype=staticmethod(lambda a, *args, **kw: a.astype(*args, **kw)),
        getitem=staticmethod(lambda a, key: a[key]),
        index_add=staticmethod(np.add.at),
        vmap=staticmethod(_vmap),
        tensor_shape=staticmethod(lambda d: d.shape),
        tensor_size=staticmethod(lambda d: d.size),
        tensor_ndim=staticmethod(lambda d: d.ndim),
        tensor_dtype=staticmethod(lambda d: d.dtype),
        tensor_item=staticmethod(lambda d: d.item()),
        repr=staticmethod(lambda d: d.__repr__()),
        len=staticmethod(lambda d: d.__len__()),
        array_interface=staticmethod(lambda d: d.__array_interface__),
        array=staticmethod(_array),
        dtype=np.dtype,
        bool=np.bool_,
        nan=np.nan,
        as_numpy=staticmethod(lambda a: a),
        _synchronize=staticmethod(lambda: None),
        _materialize=staticmethod(lambda a: a),
    )
    return type("NumpyOracleTable", (), ns)


NumpyOracleTable = _build()


def public_names():
    return [k for k in vars(NumpyOracleTable) if not k.startswith("_")]
