"""The backend function table for MI355X — name-for-name the surface the
reference's NumPy backend exports (reference: minidiff/backend/numpy.py:14-206,
114 public names; abstract stubs minidiff/backend/__init__.py:88-752), with
every array function served by libmdhip kernels through
:mod:`minidiff_amd.ndarray`.

Two consumers:
  * :mod:`minidiff_amd.tape` — this repo's tape, which drives the kernels on
    the GPU box (the reference's Python cannot travel there);
  * :mod:`minidiff_amd.plugin` — wraps this table in a ``minidiff.backend.Backend``
    subclass so the *unmodified* reference selects it with ``--backend``.

Random draws and ``.npy`` IO are produced by NumPy on the host and uploaded
(the reference aliases ``np.random.*`` — numpy.py:131-137 — so the stream of
numbers for a given seed must be NumPy's); no arithmetic on the path runs on
the CPU.
"""
from __future__ import annotations

import numpy as np

from . import ndarray as nd
from .ndarray import DeviceArray


def _host(x):
    if isinstance(x, DeviceArray):
        return x.get()
    if isinstance(x, (list, tuple)):
        return type(x)(_host(v) for v in x)
    return x


def _upload_random(fn):
    def wrapped(*args, **kwargs):
        out = fn(*[_host(a) for a in args], **{k: _host(v) for k, v in kwargs.items()})
        return nd.asarray(np.asarray(out))

    wrapped.__name__ = getattr(fn, "__name__", "random")
    return staticmethod(wrapped)


class HipBackendTable:
    tensor_constructor = staticmethod(nd.array)
    tensor_class = DeviceArray

    # ---- op functions (numpy.py:19-95) ------------------------------------
    absolute = staticmethod(nd.absolute)
    all = staticmethod(nd.all)
    any = staticmethod(nd.any)
    argmax = staticmethod(nd.argmax)
    argmin = staticmethod(nd.argmin)
    argwhere = staticmethod(nd.argwhere)
    atleast_1d = staticmethod(nd.atleast_1d)
    atleast_2d = staticmethod(nd.atleast_2d)
    atleast_3d = staticmethod(nd.atleast_3d)
    ceil = staticmethod(nd.ceil)
    copy = staticmethod(nd.copy)
    cos = staticmethod(nd.cos)
    cosh = staticmethod(nd.cosh)
    exp = staticmethod(nd.exp)
    flatten = staticmethod(nd.flatten)
    flip = staticmethod(nd.flip)
    floor = staticmethod(nd.floor)
    invert = staticmethod(nd.invert)
    log = staticmethod(nd.log)
    logical_not = staticmethod(nd.logical_not)
    max = staticmethod(nd.max)
    mean = staticmethod(nd.mean)
    min = staticmethod(nd.min)
    prod = staticmethod(nd.prod)
    ravel = staticmethod(nd.ravel)
    sign = staticmethod(nd.sign)
    sin = staticmethod(nd.sin)
    sinh = staticmethod(nd.sinh)
    squeeze = staticmethod(nd.squeeze)
    std = staticmethod(nd.std)
    sum = staticmethod(nd.sum)
    tan = staticmethod(nd.tan)
    tanh = staticmethod(nd.tanh)
    transpose = staticmethod(nd.transpose)
    add = staticmethod(nd.add)
    astype = staticmethod(nd.astype)
    broadcast_to = staticmethod(nd.broadcast_to)
    dot = staticmethod(nd.dot)
    equal = staticmethod(nd.equal)
    expand_dims = staticmethod(nd.expand_dims)
    floor_divide = staticmethod(nd.floor_divide)
    getitem = staticmethod(nd.getitem)
    greater = staticmethod(nd.greater)
    greater_equal = staticmethod(nd.greater_equal)
    less = staticmethod(nd.less)
    less_equal = staticmethod(nd.less_equal)
    logical_and = staticmethod(nd.logical_and)
    logical_or = staticmethod(nd.logical_or)
    logical_xor = staticmethod(nd.logical_xor)
    matmul = staticmethod(nd.matmul)
    mod = staticmethod(nd.mod)
    multiply = staticmethod(nd.multiply)
    not_equal = staticmethod(nd.not_equal)
    power = staticmethod(nd.power)
    reshape = staticmethod(nd.reshape)
    subtract = staticmethod(nd.subtract)
    tensordot = staticmethod(nd.tensordot)
    true_divide = staticmethod(nd.true_divide)
    clip = staticmethod(nd.clip)
    swapaxes = staticmethod(nd.swapaxes)
    where = staticmethod(nd.where)

    # ---- tensor functions (numpy.py:98-138) --------------------------------
    ones_like = staticmethod(nd.ones_like)
    ones = staticmethod(nd.ones)
    zeros_like = staticmethod(nd.zeros_like)
    zeros = staticmethod(nd.zeros)
    full_like = staticmethod(nd.full_like)
    full = staticmethod(nd.full)
    concatenate = staticmethod(nd.concatenate)
    index_add = staticmethod(nd.index_add)
    isin = staticmethod(nd.isin)
    unravel_index = staticmethod(nd.unravel_index)
    take_along_axis = staticmethod(nd.take_along_axis)

    @staticmethod
    def vmap(fun):
        # map `fun` over axis 0 (numpy.py:110-122 does it with apply_along_axis);
        # each slice is a device view, results are stacked on the device
        def mapped(arr):
            arr = nd.asarray(arr)
            outs = [nd.asarray(fun(arr[i])) for i in range(arr.shape[0])]
            return nd.stack(outs, axis=0)

        return mapped

    put_along_axis = staticmethod(nd.put_along_axis)
    repeat = staticmethod(nd.repeat)
    tile = staticmethod(nd.tile)
    arange = staticmethod(nd.arange)
    stack = staticmethod(nd.stack)

    @staticmethod
    def save(file, arr, **kw):
        np.save(file, _host(arr), **kw)

    @staticmethod
    def load(file, **kw):
        return nd.asarray(np.load(file, **kw))

    choice = _upload_random(np.random.choice)
    rand = _upload_random(np.random.rand)
    randint = _upload_random(np.random.randint)
    randn = _upload_random(np.random.randn)
    binomial = _upload_random(np.random.binomial)
    permutation = _upload_random(np.random.permutation)

    @staticmethod
    def shuffle(x):
        host = x.get()
        np.random.shuffle(host)
        nd._copy_into(x, nd.asarray(host))

    split = staticmethod(nd.split)

    # ---- tensor properties (numpy.py:141-185) -------------------------------
    @staticmethod
    def tensor_shape(data):
        return data.shape

    @staticmethod
    def tensor_size(data):
        return data.size

    @staticmethod
    def tensor_ndim(data):
        return data.ndim

    @staticmethod
    def tensor_dtype(data):
        return data.dtype

    @staticmethod
    def tensor_item(data):
        return data.item()

    @staticmethod
    def repr(data):
        return data.__repr__()

    @staticmethod
    def len(data):
        return data.__len__()

    @staticmethod
    def array_interface(data):
        raise AttributeError("device memory has no host __array_interface__; use as_numpy()")

    @staticmethod
    def array(data, dtype=None, copy=None):
        if dtype is not None and np.dtype(dtype) != data.dtype:
            if copy is False:
                raise ValueError("attempted cast, but copies are not permitted")
            return data.get().astype(dtype)
        return data.get()

    # ---- dtypes (numpy.py:188-200) ------------------------------------------
    dtype = np.dtype
    float64 = np.float64
    float32 = np.float32
    float16 = np.float16
    uint64 = np.uint64
    uint32 = np.uint32
    uint16 = np.uint16
    uint8 = np.uint8
    int64 = np.int64
    int32 = np.int32
    int16 = np.int16
    int8 = np.int8
    bool = np.bool_

    nan = np.nan

    @staticmethod
    def as_numpy(a):
        return a.get() if isinstance(a, DeviceArray) else np.asarray(a)

    # ---- extras the harness uses (not part of the reference table) -----------
    _synchronize = staticmethod(nd.synchronize)
    _materialize = staticmethod(nd.materialize)
    _materialize_many = staticmethod(nd.materialize_many)


def public_names() -> list:
    return [k for k in vars(HipBackendTable) if not k.startswith("_")]
