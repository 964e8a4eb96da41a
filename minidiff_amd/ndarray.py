"""DeviceArray — the ``tensor_class`` of the MI355X backend.

An N-d strided view (shape, element strides, offset, dtype) over a refcounted
block of HBM owned by libmdhip's caching allocator. It is the counterpart of
``numpy.ndarray`` / ``cupy.ndarray`` at the reference's backend boundary
(reference: minidiff/backend/numpy.py:15-16, cupy.py:42-44) and carries the
*implicit* surface minidiff's Tensor relies on: ``.astype``,
in-place arithmetic dunders (minidiff/tensor.py:269-362), ``__setitem__`` with
index-array keys (tensor.py:376-379), ``.size``, ``__array__``.

All arithmetic happens on the device through the C-ABI; this file only decides
*what* to launch: NumPy type resolution (``ufunc.resolve_dtypes`` — NEP 50 weak
Python scalars included), broadcasting, view construction and index plans.
"""
from __future__ import annotations

import builtins as _bi
import ctypes as C
import math
import operator
import os
import weakref
from builtins import bool as py_bool

# this module defines sum/max/min/any/all with NumPy meaning; keep the builtins
builtins_min, builtins_max, builtins_sum, builtins_any, builtins_all = _bi.min, _bi.max, _bi.sum, _bi.any, _bi.all

import numpy as np

from . import _capi
from . import lazy as _lz
from . import narrow as _narrow
from ._capi import ArrayDesc, IndexPlan, MAX_NDIM

# The C route of eager calls (csrc/fastpath.c): block owner, array fields and the float elementwise entries in C. Off with
# MDHIP_FASTPATH=0 (A/B of the host cost, scripts/host_overhead.py) and under MDHIP_TRACE (the call log sits in the ctypes layer).
if os.environ.get("MDHIP_FASTPATH", "1") != "0" and not os.environ.get("MDHIP_TRACE"):
    try:
        if os.environ.get("MDHIP_FASTPATH_SO"):   # TESTS ONLY: a named build of the extension (the sanitizer build, tests/test_sanitized_double.py)
            import importlib.util
            import sys as _sys
            _spec = importlib.util.spec_from_file_location(__package__ + "._fastpath", os.environ["MDHIP_FASTPATH_SO"])
            _fp = importlib.util.module_from_spec(_spec)
            _spec.loader.exec_module(_fp)
            _sys.modules[__package__ + "._fastpath"] = _fp
        else:
            from . import _fastpath as _fp
    except ImportError as e:
        # Built by `make -C minidiff_amd/csrc` (__graft_entry__.build()) next to libmdhip.so. NEVER built here: under an N-rank
        # launch every rank imports at once and N compilers would write the same file. Without it the eager host route is the
        # Python implementation the extension fronts — same results, ~3 us more per call.
        import sys as _sys
        print(f"[mdhip] minidiff_amd/_fastpath is not built ({e}); using the Python host route "
              "(build: python -c 'import __graft_entry__ as g; g.build()')", file=_sys.stderr)
        _fp = None
    if _fp is not None and _fp.ABI_DESC_BYTES != C.sizeof(ArrayDesc):
        raise ImportError("minidiff_amd/_fastpath was built against another include/mdhip.h: rebuild (make -C minidiff_amd/csrc)")
else:
    _fp = None

# Lazy fusion of elementwise chains (minidiff_amd/lazy.py) is opt-in; eager —
# one kernel per backend call, the reference's execution model — is the default.
_LAZY = os.environ.get("MDHIP_LAZY", "0") == "1"
# lazy mode: results with fewer elements than this are computed eagerly all the same (a fused program saves HBM passes; below ~10^4
# elements there are none to save and building the program costs more host time than the kernels it replaces). 0: fuse everything.
_LAZY_MIN = int(os.environ.get("MDHIP_LAZY_MIN", "0"))


def set_lazy(flag: bool) -> bool:
    """Switch lazy fusion on/off; returns the previous setting."""
    global _LAZY
    prev, _LAZY = _LAZY, py_bool(flag)
    if _fp is not None:
        _fp.set_lazy(_LAZY)
    return prev


def lazy_enabled() -> bool:
    return _LAZY


# launches issued for pending expressions (tests assert that fusion really happened)
FUSION_STATS = {"vm_eval": 0, "vm_reduce": 0, "vm_eval_multi": 0, "vm_eval_reduce_cols": 0, "deferred_cols": 0, "gemm_deferred": 0, "gemm_epilogue": 0}

_DTYPE_CODES = {
    np.dtype(np.bool_): _capi.BOOL,
    np.dtype(np.int32): _capi.I32,
    np.dtype(np.int64): _capi.I64,
    np.dtype(np.float32): _capi.F32,
    np.dtype(np.float64): _capi.F64,
    # storage-only dtypes (reference table numpy.py:188-200): held, moved and converted on the device; arithmetic on them is
    # promote -> wide kernel -> demote (minidiff_amd/narrow.py). Any compute entry point handed one of these codes refuses it.
    np.dtype(np.int8): _capi.I8, np.dtype(np.int16): _capi.I16,
    np.dtype(np.uint8): _capi.U8, np.dtype(np.uint16): _capi.U16, np.dtype(np.uint32): _capi.U32, np.dtype(np.uint64): _capi.U64,
    np.dtype(np.float16): _capi.F16,
}
_FLOAT_CODES = (_capi.F32, _capi.F64)
_NARROW_MIN = _capi.I8   # codes >= this: storage-only


def dtype_code(dt) -> int:
    try:
        return _DTYPE_CODES[dt]
    except KeyError:
        pass
    dt = np.dtype(dt)
    code = _DTYPE_CODES.get(dt)
    if code is None:
        raise TypeError(f"dtype {dt} is not supported by the MI355X backend (bool, int8/16/32/64, uint8/16/32/64, float16/32/64)")
    return code


def _lib() -> _capi.Library:
    lib = _capi._LIB
    if lib is None:
        lib = _capi.load()
    return lib


class _PyBuffer:
    """Owner of one allocator block; freed when the last view drops it. (Pure-Python form; `_fastpath.Buffer` is the same
    object in C and the one in use unless MDHIP_FASTPATH=0 / MDHIP_TRACE.)"""

    __slots__ = ("ptr", "nbytes", "_free", "deps", "task", "__weakref__")

    def __init__(self, nbytes: int):
        self.deps = None  # {id: weakref} of pending (lazy) arrays that read this block
        self.task = None  # deferred fill of this block (lazy mode: _ColsTask); run before any access to its bytes
        lib = _lib()
        p = C.c_void_p()
        lib.alloc(builtins_max(int(nbytes), 1), C.byref(p))
        self.ptr = p.value or 0
        self.nbytes = int(nbytes)
        self._free = lib.cdll.mdhip_free

    def __del__(self):
        try:
            if self.ptr:
                self._free(C.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:  # interpreter teardown
            pass


_Buffer = _fp.Buffer if _fp is not None else _PyBuffer

_PINNED_MIN = 1 << 20


class _PinnedOwner:
    """Returns a pinned host block to the pool when the ctypes view (hence every ndarray over it) is gone."""

    __slots__ = ("ptr", "_free")

    def __init__(self, ptr, free):
        self.ptr, self._free = ptr, free

    def __del__(self):
        try:
            self._free(C.c_void_p(self.ptr))
        except Exception:  # interpreter teardown
            pass


def _pinned_empty(shape, dtype, nbytes):
    lib = _lib()
    p = C.c_void_p()
    try:
        lib.host_alloc(nbytes, C.byref(p))
    except MemoryError:
        return None
    view = (C.c_char * nbytes).from_address(p.value)
    view._owner = _PinnedOwner(p.value, lib.cdll.mdhip_host_free)
    return np.frombuffer(view, dtype=dtype).reshape(shape)


_C_STRIDES: dict = {}
_DESC_TEMPLATES: dict = {}


def _c_strides(shape) -> tuple:
    st = _C_STRIDES.get(shape)           # (the same few shapes recur in every sweep: a dict hit instead of a loop per result array)
    if st is None:
        acc, out = 1, [0] * len(shape)
        for i in range(len(shape) - 1, -1, -1):
            out[i] = acc
            acc *= shape[i]
        st = tuple(out)
        if len(_C_STRIDES) < 4096:
            _C_STRIDES[shape] = st
    return st


def _prod(shape) -> int:
    n = 1
    for s in shape:
        n *= s
    return n


def _normalize_shape(shape) -> tuple:
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(operator.index(s)) for s in shape)


def _axis_error(axis, ndim):
    return np.exceptions.AxisError(axis, ndim)


def normalize_axis(axis, ndim) -> int:
    ax = operator.index(axis)
    if ax < -ndim or ax >= ndim:
        raise _axis_error(axis, ndim)
    return ax + ndim if ax < 0 else ax


def _comparable(o) -> py_bool:
    """`array == o` for an `o` that is no number and no array of numbers (None, a string, an arbitrary object) is all False in
    NumPy (elementwise comparison with an object), not an error."""
    if o is None or isinstance(o, (str, bytes)):
        return False
    return True


def normalize_axes(axis, ndim) -> tuple:
    if axis is None:
        return tuple(range(ndim))
    if isinstance(axis, (tuple, list)):
        out = tuple(normalize_axis(a, ndim) for a in axis)
        if len(set(out)) != len(out):
            raise ValueError("duplicate value in 'axis'")
        return out
    return (normalize_axis(axis, ndim),)


class DeviceArray(_fp.ArrayBase if _fp is not None else object):
    # fields (a C struct in _fastpath.ArrayBase, slots otherwise): _buf, _offset, shape, _strides, dtype, _code,
    #   _expr        pending expression (lazy mode); _buf is None until materialised
    #   _cdesc       own-shape C descriptor, built once (geometry and block never change)
    #   _tasks       weakrefs to deferred reductions of this pending expression (_ColsTask)
    #   _dependents  pending arrays that took THIS pending product as a leaf before it had a block
    __slots__ = () if _fp is not None else ("_buf", "_offset", "shape", "_strides", "dtype", "_code", "_expr", "_cdesc", "_tasks", "_dependents", "__weakref__")
    __array_priority__ = 1000.0
    __hash__ = None

    if _fp is None:
        def __init__(self, buf: _Buffer, offset: int, shape: tuple, strides: tuple, dtype: np.dtype, code: int = -1):
            self._buf = buf
            self._offset = offset
            self.shape = shape
            self._strides = strides
            self.dtype = dtype
            self._code = code if code >= 0 else dtype_code(dtype)
            self._expr = None
            self._cdesc = None
            self._tasks = None
            self._dependents = None

    # ---- lazy evaluation -----------------------------------------------------
    @staticmethod
    def _pending(expr, shape, dtype) -> "DeviceArray":
        arr = DeviceArray(None, 0, shape, _c_strides(shape), dtype)
        arr._expr = expr
        expr.owner = weakref.ref(arr)
        for leaf in expr.leaves.values():
            b = leaf._buf
            if b is None:  # a deferred matrix product: it registers this reader once it has a block
                if leaf._dependents is None:
                    leaf._dependents = []
                leaf._dependents.append(weakref.ref(arr))
                continue
            _register_reader(b, arr)
        return arr

    @staticmethod
    def _pending_gemm(a, b, shape, dtype, cdt) -> "DeviceArray":
        """a @ b, not launched yet (lazy mode). Operands are concrete; in-place writes into them flush the product first."""
        arr = DeviceArray(None, 0, shape, _c_strides(shape), dtype)
        arr._expr = _lz.gemm(a, b, cdt)
        arr._expr.owner = weakref.ref(arr)
        for x in (a, b):
            _register_reader(x._buf, arr)
        return arr

    def _materialize(self):
        e = self._expr
        if e is None:
            return
        if e.kind == _lz.GEMM:
            a, b = e.args
            self._buf = _Buffer(_prod(self.shape) * self.dtype.itemsize)
            self._expr = None
            try:
                _gemm_into(a, b, self)
            except BaseException:
                self._buf, self._expr, self._cdesc = None, e, None
                raise
            FUSION_STATS["gemm_deferred"] += 1
            deps, self._dependents = self._dependents, None
            for r in deps or ():
                d = r()
                if d is not None and d._expr is not None:
                    _register_reader(self._buf, d)
            return
        prog, keep = _lz.build_program(e, self.shape)
        self._buf = _Buffer(_prod(self.shape) * self.dtype.itemsize)
        self._expr = None
        saved_tasks, claimed = self._tasks, []
        try:
            tasks = self._live_tasks()
            if self.size:
                fused = None
                if tasks:
                    # a reduce-to-shape of this expression is still owed: evaluate and reduce in ONE pass
                    t = tasks[0]
                    res = t.res
                    t.claim()
                    claimed.append(t)
                    try:
                        _lib().vm_eval_reduce_cols(prog, t.code, self.desc(), res.desc())
                        FUSION_STATS["vm_eval_reduce_cols"] += 1
                        fused = t
                    except ValueError:  # shape not covered by the one-pass kernel
                        pass
                if fused is None:
                    _lib().vm_eval(prog, self.desc())
                    FUSION_STATS["vm_eval"] += 1
                for k, t in enumerate(tasks):
                    if t is not fused:  # from the materialised value (one more read, same result)
                        r = res if k == 0 else t.res
                        if t not in claimed:
                            t.claim()
                            claimed.append(t)
                        _lib().reduce(t.code, self.desc(), r.desc(), 1)
        except BaseException:
            # the launch did not happen: the array must not look materialised over an unwritten block, and the column sums
            # that were claimed for this pass are owed again (their result blocks are still unfilled)
            self._buf, self._expr, self._cdesc = None, e, None
            for t in claimed:
                t.unclaim(self)
            self._tasks = saved_tasks
            raise
        del keep

    def _live_tasks(self):
        ts, self._tasks = self._tasks, None
        if not ts:
            return ()
        out = []
        for r in ts:
            t = r()
            if t is not None and not t.done and t._buf() is not None:
                out.append(t)
        return out

    def materialize(self) -> "DeviceArray":
        if self._buf is None:
            self._materialize()
        return self

    # ---- construction -------------------------------------------------------
    @staticmethod
    def empty(shape, dtype) -> "DeviceArray":
        shape = _normalize_shape(shape)
        if len(shape) > MAX_NDIM:
            raise ValueError(f"maximum supported dimension for a DeviceArray is {MAX_NDIM}, found {len(shape)}")
        for s in shape:
            if s < 0:
                raise ValueError("negative dimensions are not allowed")
        dtype = np.dtype(dtype)
        code = dtype_code(dtype)
        buf = _Buffer(_prod(shape) * dtype.itemsize)
        return DeviceArray(buf, 0, shape, _c_strides(shape), dtype, code)

    @staticmethod
    def _new(shape: tuple, dtype: np.dtype) -> "DeviceArray":
        """Result buffer of an op: `shape` is already a validated tuple of ints, `dtype` an np.dtype
        (skips the argument normalisation of `empty`; the dispatch cost of a small op is Python)."""
        if _fp is not None:
            return _fp.new_array(shape, dtype, dtype_code(dtype))
        n = dtype.itemsize
        for s in shape:
            n *= s
        return DeviceArray(_Buffer(n), 0, shape, _c_strides(shape), dtype, dtype_code(dtype))

    @staticmethod
    def from_numpy(arr) -> "DeviceArray":
        arr = np.asarray(arr)
        dtype_code(arr.dtype)
        host = arr if arr.flags.c_contiguous else np.array(arr, order="C")  # (ascontiguousarray would promote 0-d to 1-d)
        out = DeviceArray.empty(host.shape, host.dtype)
        if host.size == 1 and out._code < _NARROW_MIN:
            # one element (a wrapped Python scalar): a fill launch instead of an upload — no stream
            # synchronisation, and it stays capturable into a graph (graph.py)
            _fill(out, host.reshape(()).item())
        elif host.nbytes:
            _lib().h2d(out.ptr, host.ctypes.data, host.nbytes)
        return out

    # ---- geometry -----------------------------------------------------------
    def _block(self) -> _Buffer:
        """The allocator block with its bytes valid: evaluates a pending expression / runs a deferred fill."""
        b = self._buf
        if b is None:
            self._materialize()
            b = self._buf
        if b.task is not None:
            b.task.run()
        return b

    @property
    def ptr(self) -> int:
        return self._block().ptr + self._offset * self.dtype.itemsize

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def size(self) -> int:
        return _prod(self.shape)

    @property
    def itemsize(self) -> int:
        return self.dtype.itemsize

    @property
    def nbytes(self) -> int:
        return self.size * self.dtype.itemsize

    @property
    def strides(self) -> tuple:  # bytes, like numpy
        return tuple(s * self.dtype.itemsize for s in self._strides)

    @property
    def is_c_contiguous(self) -> py_bool:
        if 0 in self.shape:
            return True      # (an empty array is contiguous both ways, as in NumPy)
        acc = 1
        for n, s in zip(reversed(self.shape), reversed(self._strides)):
            if n == 1:
                continue
            if n == 0:
                return True
            if s != acc:
                return False
            acc *= n
        return True

    @property
    def base(self):
        return self._block()

    @property
    def T(self) -> "DeviceArray":
        return transpose(self)

    def _view(self, offset, shape, strides) -> "DeviceArray":
        if self._buf is None:
            self._materialize()
        return DeviceArray(self._buf, offset, tuple(shape), tuple(strides), self.dtype, self._code)

    def desc(self, shape=None) -> ArrayDesc:
        """C descriptor; with `shape`, broadcast (stride 0) to that shape."""
        b = self._buf
        if b is not None and b.task is not None:
            b.task.run()
        if shape is None or shape == self.shape:
            d = self._cdesc
            if d is None:
                key = (self.shape, self._strides, self._code)
                tmpl = _DESC_TEMPLATES.get(key)      # geometry part of the descriptor: built once per (shape, strides, dtype), then copied
                if tmpl is None:
                    t = ArrayDesc()
                    t.dtype = self._code
                    nd = len(self.shape)
                    t.ndim = nd
                    if nd:
                        t.shape[:nd] = self.shape
                        t.strides[:nd] = self._strides
                    tmpl = bytes(t)
                    if len(_DESC_TEMPLATES) < 4096:
                        _DESC_TEMPLATES[key] = tmpl
                d = ArrayDesc.from_buffer_copy(tmpl)
                d.data = self.ptr
                self._cdesc = d
            return d
        d = ArrayDesc()
        d.data = self.ptr
        d.dtype = self._code
        nd = len(shape)
        lead = nd - len(self.shape)
        st = [0] * nd
        for i, (n, s) in enumerate(zip(self.shape, self._strides)):
            tgt = shape[lead + i]
            if n == tgt:
                st[lead + i] = s
            elif n != 1:
                raise ValueError(f"operands could not be broadcast together with shapes {self.shape} {shape}")
        d.ndim = nd
        if nd:
            d.shape[:nd] = shape
            d.strides[:nd] = st
        return d

    # ---- host transfer ------------------------------------------------------
    def get(self) -> np.ndarray:
        """D2H copy (synchronises the stream). Large results land in a page-locked block from
        libmdhip's host pool (returned to the pool when the last NumPy view of it dies): a copy into
        fresh pageable memory runs at the page-fault rate, not the link rate."""
        src = self if self.is_c_contiguous else _ccopy(self)
        nbytes = src.size * self.dtype.itemsize
        host = None
        if nbytes >= _PINNED_MIN:
            host = _pinned_empty(self.shape, self.dtype, nbytes)
        if host is None:
            host = np.empty(self.shape, dtype=self.dtype)
        if nbytes:
            _lib().d2h(host.ctypes.data, src.ptr, nbytes)
        return host

    def __array__(self, dtype=None, copy=None):
        host = self.get()
        if dtype is not None and np.dtype(dtype) != host.dtype:
            host = host.astype(dtype)
        return host

    @property
    def __cuda_array_interface__(self):
        return {
            "shape": self.shape,
            "typestr": self.dtype.str,
            "data": (self.ptr, False),
            "version": 3,
            "strides": None if self.is_c_contiguous else self.strides,
        }

    def item(self):
        if self.size != 1:
            raise ValueError("can only convert an array of size 1 to a Python scalar")
        return self.get().reshape(()).item()

    def tolist(self):
        return self.get().tolist()

    def __repr__(self):
        return repr(self.get()).replace("array(", "DeviceArray(", 1)

    def __str__(self):
        return str(self.get())

    def __format__(self, spec):
        return format(self.get()[()] if self.ndim == 0 else self.get(), spec)

    def __round__(self, ndigits=None):
        return round(self._scalar(), ndigits) if ndigits is not None else round(self._scalar())

    def _scalar(self):
        if self.size != 1:
            raise TypeError("only length-1 arrays can be converted to Python scalars")
        return self.item()

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    def __bool__(self):
        if self.size != 1:
            raise ValueError("The truth value of an array with more than one element is ambiguous. Use a.any() or a.all()")
        return py_bool(self.item())

    def __float__(self):
        return float(self._scalar())

    def __int__(self):
        return int(self._scalar())

    def __complex__(self):
        return complex(self._scalar())

    def __index__(self):
        if self.size != 1 or self.dtype.kind not in "iu":
            raise TypeError("only integer scalar arrays can be converted to a scalar index")
        return int(self.item())

    def __iter__(self):
        if not self.shape:
            raise TypeError("iteration over a 0-d array")
        for i in range(self.shape[0]):
            yield self[i]

    # ---- methods the reference calls on raw arrays ---------------------------
    def astype(self, dtype, copy=True, **_):
        return astype(self, dtype, copy=copy)

    def copy(self, order="C"):
        return copy(self, order=order)

    def reshape(self, *shape, order="C"):
        if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)):
            shape = shape[0]
        return reshape(self, shape, order=order)

    def transpose(self, *axes):
        if len(axes) == 1 and (axes[0] is None or isinstance(axes[0], (tuple, list))):
            axes = axes[0]
        elif len(axes) == 0:
            axes = None
        return transpose(self, axes)

    def ravel(self, order="C"):
        return ravel(self, order=order)

    def flatten(self, order="C"):
        return flatten(self, order=order)

    def squeeze(self, axis=None):
        return squeeze(self, axis)

    def sum(self, axis=None, dtype=None, out=None, keepdims=False):
        return sum(self, axis=axis, dtype=dtype, out=out, keepdims=keepdims)

    def mean(self, axis=None, dtype=None, out=None, keepdims=False):
        return mean(self, axis=axis, dtype=dtype, out=out, keepdims=keepdims)

    def max(self, axis=None, out=None, keepdims=False):
        return max(self, axis=axis, out=out, keepdims=keepdims)

    def min(self, axis=None, out=None, keepdims=False):
        return min(self, axis=axis, out=out, keepdims=keepdims)

    def prod(self, axis=None, dtype=None, out=None, keepdims=False):
        return prod(self, axis=axis, dtype=dtype, out=out, keepdims=keepdims)

    def any(self, axis=None, out=None, keepdims=False):
        return any(self, axis=axis, out=out, keepdims=keepdims)

    def all(self, axis=None, out=None, keepdims=False):
        return all(self, axis=axis, out=out, keepdims=keepdims)

    def argmax(self, axis=None, out=None, keepdims=False):
        return argmax(self, axis=axis, out=out, keepdims=keepdims)
    def clip(self, a_min=None, a_max=None, **kw): return clip(self, a_min, a_max, **kw)
    def std(self, axis=None, dtype=None, out=None, ddof=0, keepdims=False): return std(self, axis=axis, dtype=dtype, out=out, ddof=ddof, keepdims=keepdims)
    def swapaxes(self, a0, a1): return swapaxes(self, a0, a1)
    def nonzero(self): return nonzero(self)
    def repeat(self, repeats, axis=None): return repeat(self, repeats, axis=axis)


    def argmin(self, axis=None, out=None, keepdims=False):
        return argmin(self, axis=axis, out=out, keepdims=keepdims)

    def dot(self, other):
        return dot(self, other)

    def fill(self, value):
        _fill(self, value)

    # ---- operators ----------------------------------------------------------
    def __add__(self, o): return add(self, o)
    def __radd__(self, o): return add(o, self)
    def __sub__(self, o): return subtract(self, o)
    def __rsub__(self, o): return subtract(o, self)
    def __mul__(self, o): return multiply(self, o)
    def __rmul__(self, o): return multiply(o, self)
    def __truediv__(self, o): return true_divide(self, o)
    def __rtruediv__(self, o): return true_divide(o, self)
    def __floordiv__(self, o): return floor_divide(self, o)
    def __rfloordiv__(self, o): return floor_divide(o, self)
    def __mod__(self, o): return mod(self, o)
    def __divmod__(self, o): return floor_divide(self, o), mod(self, o)
    def __rdivmod__(self, o): return floor_divide(o, self), mod(o, self)
    def __contains__(self, v): return py_bool(any(equal(self, v)).item())
    def __rmod__(self, o): return mod(o, self)
    def __pow__(self, o): return power(self, o)
    def __rpow__(self, o): return power(o, self)
    def __matmul__(self, o): return matmul(self, o)
    def __rmatmul__(self, o): return matmul(o, self)
    def __neg__(self): return negative(self)
    def __pos__(self): return _ccopy(self)
    def __abs__(self): return absolute(self)
    def __invert__(self): return invert(self)
    def __lt__(self, o): return less(self, o)
    def __le__(self, o): return less_equal(self, o)
    def __gt__(self, o): return greater(self, o)
    def __ge__(self, o): return greater_equal(self, o)
    def __eq__(self, o): return equal(self, o) if _comparable(o) else full(self.shape, False, dtype=np.bool_)
    def __ne__(self, o): return not_equal(self, o) if _comparable(o) else full(self.shape, True, dtype=np.bool_)
    def __and__(self, o): return logical_and(self, o) if self.dtype == np.bool_ else NotImplemented
    def __or__(self, o): return logical_or(self, o) if self.dtype == np.bool_ else NotImplemented
    def __xor__(self, o): return logical_xor(self, o) if self.dtype == np.bool_ else NotImplemented

    # in-place forms act on this array's memory (minidiff/tensor.py:269-362)
    def __iadd__(self, o): return _inplace(np.add, _capi.B_ADD, add, self, o)
    def __isub__(self, o): return _inplace(np.subtract, _capi.B_SUB, subtract, self, o)
    def __imul__(self, o): return _inplace(np.multiply, _capi.B_MUL, multiply, self, o)
    def __itruediv__(self, o): return _inplace(np.true_divide, _capi.B_TRUE_DIV, true_divide, self, o)
    def __ifloordiv__(self, o): return _inplace(np.floor_divide, _capi.B_FLOOR_DIV, floor_divide, self, o)
    def __imod__(self, o): return _inplace(np.remainder, _capi.B_MOD, mod, self, o)
    def __ipow__(self, o): return _inplace(np.power, _capi.B_POW, power, self, o)

    def __imatmul__(self, o):
        if _any_narrow((self, o), None):
            return _narrow_inplace(np.matmul, matmul, self, o)
        res = matmul(self, o)
        if res.shape != self.shape:
            raise ValueError(f"inplace matrix multiplication requires the result shape {res.shape} to equal {self.shape}")
        _copy_into(self, res)
        return self

    def __getitem__(self, key):
        return getitem(self, key)

    def __setitem__(self, key, value):
        setitem(self, key, value)


def _inplace(ufunc, code, fn, a, b):
    """a OP= b: the in-place kernel (any of the twelve dtypes: one launch; a loop whose result dtype differs from a's — uint8 += int16
    under same-kind casting — computes into a temporary and converts)."""
    return _binary(ufunc, code, a, b, out=a)


# =============================================================================
# operands and type resolution
# =============================================================================
_WEAK = (py_bool, int, float)


def _scalar_desc(value, code: int) -> ArrayDesc:
    d = ArrayDesc()
    d.dtype = code
    d.is_scalar = 1
    if code in _FLOAT_CODES:
        d.scalar_f = float(value)
    else:
        v = int(value)
        if not (_INT64_MIN <= v < (1 << 64)):
            raise OverflowError(f"Python integer {v} does not fit a 64-bit scalar operand")      # (backstop: callers range-check first)
        if v > _INT64_MAX:      # a uint64 value >= 2**63 travels as its bits, under the uint64 code (md_scalar_as reads it unsigned)
            d.dtype = _capi.U64
            v -= 1 << 64
        d.scalar_i = v
    return d


def asarray(obj, dtype=None) -> DeviceArray:
    if isinstance(obj, DeviceArray):
        if dtype is not None and np.dtype(dtype) != obj.dtype:
            return astype(obj, dtype)
        return obj
    if isinstance(obj, (list, tuple)) and _contains_device(obj):
        obj = _to_host_nested(obj)
    arr = np.array(obj, dtype=dtype) if not isinstance(obj, np.ndarray) else (obj if dtype is None else obj.astype(dtype))
    if arr.dtype.kind in "OUSV":
        raise TypeError(f"cannot place an object of dtype {arr.dtype} on the device")
    return DeviceArray.from_numpy(arr)


def _contains_device(seq) -> py_bool:
    for x in seq:
        if isinstance(x, DeviceArray):
            return True
        if isinstance(x, (list, tuple)) and _contains_device(x):
            return True
    return False


def _to_host_nested(seq):
    out = []
    for x in seq:
        if isinstance(x, DeviceArray):
            out.append(x.get())
        elif isinstance(x, (list, tuple)):
            out.append(_to_host_nested(x))
        else:
            out.append(x)
    return out


def array(obj, dtype=None, copy=True, **kw) -> DeviceArray:
    _defaults_only("array", kw)
    """tensor_constructor (reference: numpy.py:15 ``np.array``)."""
    if isinstance(obj, DeviceArray):
        if dtype is not None and np.dtype(dtype) != obj.dtype:
            return astype(obj, dtype)
        return globals()["copy"](obj) if copy else obj
    return asarray(obj, dtype=dtype)


def _operand(x):
    """-> DeviceArray | python scalar (weak) | numpy scalar (strong)."""
    if isinstance(x, DeviceArray):
        return x
    if isinstance(x, _WEAK):
        return x
    if isinstance(x, np.generic):
        dtype_code(x.dtype)
        return x
    if x is None:
        raise TypeError("unsupported operand type(s): 'NoneType'")
    return asarray(x)


_RESOLVE_CACHE: dict = {}


def _kind_key(x):
    if isinstance(x, DeviceArray):
        return x.dtype
    if isinstance(x, np.generic):
        return x.dtype
    if isinstance(x, py_bool):
        return np.dtype(np.bool_)
    if isinstance(x, int):
        return int
    return float


def _resolve(ufunc, keys: tuple) -> tuple:
    ck = (ufunc, keys)
    hit = _RESOLVE_CACHE.get(ck)
    if hit is None:
        if builtins_all(k is int or k is float for k in keys):
            # no typed operand at all: NumPy uses the default dtypes
            keys2 = tuple(np.dtype(np.int64) if k is int else np.dtype(np.float64) for k in keys)
        else:
            keys2 = keys
        res = ufunc.resolve_dtypes(keys2 + (None,))
        for dt in res:
            dtype_code(dt)
        hit = tuple(res)
        _RESOLVE_CACHE[ck] = hit
    return hit


def _broadcast_shapes(a: tuple, b: tuple) -> tuple:
    if a == b:
        return a
    na, nb = len(a), len(b)
    n = na if na > nb else nb
    out = [1] * n
    for i in range(1, n + 1):
        x = a[-i] if i <= na else 1
        y = b[-i] if i <= nb else 1
        if x == y or y == 1:
            out[-i] = x
        elif x == 1:
            out[-i] = y
        else:
            raise ValueError(f"operands could not be broadcast together with shapes {a} {b} ")
    return tuple(out)


def _operand_desc(x, shape, code_for_scalar) -> ArrayDesc:
    if isinstance(x, DeviceArray):
        return x.desc(shape)
    if isinstance(x, np.generic):
        code = dtype_code(x.dtype)
        if code >= _NARROW_MIN:     # a storage-only NumPy scalar travels as its VALUE (the loop dtype is passed separately)
            code = _capi.F64 if x.dtype.kind == "f" else _capi.I64
        return _scalar_desc(x.item(), code)
    return _scalar_desc(x, code_for_scalar)


_INT64_MIN, _INT64_MAX = -(2 ** 63), 2 ** 63 - 1


def _scalar_code(x, loop_dt) -> int:
    """dtype code under which a weak Python scalar travels to the kernel (NEP 50: a Python int must FIT an integer loop dtype)."""
    if isinstance(x, float):
        return _capi.F64
    if isinstance(x, int) and not isinstance(x, py_bool):
        if loop_dt.kind in "iu":
            info = np.iinfo(loop_dt)
            if not (info.min <= x <= info.max):
                raise OverflowError(f"Python integer {x} out of bounds for {loop_dt}")
        elif not (_INT64_MIN <= x <= _INT64_MAX):
            if loop_dt.kind == "f":
                return _capi.F64
            raise OverflowError(f"Python integer {x} out of bounds for {loop_dt}")
    return _capi.I64


_FLOAT_DT = {np.dtype(np.float32): _capi.F32, np.dtype(np.float64): _capi.F64}


def _before_write(arr: "DeviceArray"):
    """Evaluate every pending expression that still reads `arr`'s block (called by all
    in-place entry points: lazy results must see the bytes as they were when recorded)."""
    b = arr._buf
    if b is not None and b.deps:
        pending = [r() for r in list(b.deps.values())]
        b.deps = None
        for d in pending:
            if d is not None:
                d._materialize()


def _register_reader(buf, arr):
    """`arr` (pending) reads `buf`: an in-place write into that block must evaluate it first (_before_write)."""
    if buf.deps is None:
        buf.deps = {}
    deps, key = buf.deps, id(arr)
    deps[key] = weakref.ref(arr, lambda _r, k=key, d=deps: d.pop(k, None))  # (arrays are unhashable)


def _as_expr(x, cdt):
    """operand -> expression node of a program computing in float type `cdt`."""
    if isinstance(x, DeviceArray):
        e = x._expr
        if e is not None:
            if e.kind == _lz.GEMM:
                return _lz.leaf(x, cdt)   # a deferred product is a leaf of elementwise programs
            if e.cdt == cdt:
                return e
            x._materialize()
        return _lz.leaf(x, cdt)
    if isinstance(x, np.generic):
        return _lz.const(x.item(), cdt)
    return _lz.const(x, cdt)


def _lazy_node(kind, code, operands, cdt, shape, odt):
    """Build a pending result, materialising the largest pending operand(s) while the
    program would not fit the interpreter."""
    while True:
        parts = [_as_expr(x, cdt) for x in operands]
        e = _lz.combine(kind, code, parts, cdt)
        if _lz.fits(e):
            return DeviceArray._pending(e, shape, odt)
        big = None
        for x in operands:
            if isinstance(x, DeviceArray) and x._expr is not None and x._expr.kind != _lz.GEMM and (big is None or x._expr.n > big._expr.n):
                big = x
        if big is None:
            return None
        big._materialize()


def _binary(ufunc, code, a, b, out=None):
    a = _operand(a)
    b = _operand(b)
    a_arr = isinstance(a, DeviceArray)
    b_arr = isinstance(b, DeviceArray)
    if not a_arr and not b_arr:
        a = asarray(a)
        a_arr = True
    loop = _resolve(ufunc, (_kind_key(a), _kind_key(b)))
    cdt, odt = loop[0], loop[2]
    # (a storage-only loop dtype — int8 * int8, float16 + float16, bool / bool -> float16 — or storage-only operands of a wide loop
    # go to the same entry point: ONE launch that loads each operand in its own type and computes in the loop dtype's carrier,
    # csrc/narrow.hip)
    if a_arr and b_arr:
        shape = _broadcast_shapes(a.shape, b.shape)
    else:
        shape = a.shape if a_arr else b.shape
    if code == _capi.B_POW and cdt.kind == "i" and _prod(shape) > 0:
        _check_int_power(b)      # (NumPy raises from inside its loop: an EMPTY result never meets the negative exponent)
    if _capi.B_EQ <= code <= _capi.B_GE and cdt.kind in "iu":
        # a Python int beyond the loop dtype's range compares as the number it is (NumPy 2: `u64_array >= -1` is all True,
        # `i32_array == 2**40` all False) — the answer does not depend on the array's values
        const = _oob_compare(code, a, b, cdt)
        if const is not None:
            if out is None:
                return full(shape, const, dtype=np.bool_)
            if shape != out.shape:
                raise ValueError(f"non-broadcastable output operand with shape {out.shape} doesn't match the broadcast shape {shape}")
            _copy_into(out, const)      # (bool casts safely into every dtype)
            return out
    if _LAZY and out is None and _prod(shape) >= _LAZY_MIN and not ((a_arr and a._code >= _NARROW_MIN) or (b_arr and b._code >= _NARROW_MIN)):
        # (storage-only operands are never leaves of a fused program: the interpreter and the generated kernels read the five compute dtypes)
        pcdt = _FLOAT_DT.get(cdt)
        if pcdt is None and cdt.kind == "b" and code >= _capi.B_LAND:
            # logical op on bool operands: join the program of a pending operand
            for x in (a, b):
                if isinstance(x, DeviceArray) and x._expr is not None:
                    pcdt = x._expr.cdt
                    break
        if pcdt is not None and _prod(shape) > 0:
            res = _lazy_node(_lz.BINARY, code, (a, b), pcdt, shape, odt)
            if res is not None:
                return res
    if a_arr:
        a = _straighten(a)
    if b_arr:
        b = _straighten(b)
    if out is not None:
        if a_arr:
            a = _unalias(a, out)
        if b_arr:
            b = _unalias(b, out)
    da = _operand_desc(a, shape, 0 if a_arr else _scalar_code(a, cdt))
    db = _operand_desc(b, shape, 0 if b_arr else _scalar_code(b, cdt))
    if out is None:
        res = DeviceArray._new(shape, odt)
        _lib().binary(code, da, db, res.desc(), _DTYPE_CODES[cdt])
        return res
    _before_write(out)
    # in-place: NumPy's same_kind casting rule for the `out=` operand
    if shape != out.shape:
        raise ValueError(f"non-broadcastable output operand with shape {out.shape} doesn't match the broadcast shape {shape}")
    if not np.can_cast(odt, out.dtype, casting="same_kind"):
        raise np._core._exceptions._UFuncOutputCastingError(ufunc, "same_kind", odt, out.dtype, 0)
    if odt == out.dtype:
        _lib().binary(code, da, db, out.desc(), _DTYPE_CODES[cdt])
    else:
        tmp = DeviceArray._new(shape, odt)
        _lib().binary(code, da, db, tmp.desc(), _DTYPE_CODES[cdt])
        _copy_into(out, tmp)
    return out


def _oob_compare(code, a, b, cdt):
    """The constant answer of `array CMP python_int` when the int lies outside cdt's range; None when it fits."""
    for x, swapped in ((b, False), (a, True)):
        if type(x) is int:
            info = np.iinfo(cdt)
            if info.min <= x <= info.max:
                return None
            above = x > info.max          # array < x everywhere (above) or array > x everywhere
            if swapped:                   # `x CMP array`: the array is on the right
                above = not above
            if code == _capi.B_EQ:
                return False
            if code == _capi.B_NE:
                return True
            return above if code in (_capi.B_LT, _capi.B_LE) else not above
    return None


def _value_desc(v, dt) -> ArrayDesc:
    """Scalar descriptor of a value assigned / added into an array of dtype `dt` (NEP 50: a Python int must fit `dt`)."""
    if isinstance(v, np.generic):
        v = v.item()
    elif type(v) is int:
        v = _py_int_for(dt, v)
    if isinstance(v, (float, complex)):
        return _scalar_desc(v, _capi.F64)
    return _scalar_desc(v, _capi.I64)


def _straighten(x):
    """A large operand whose LAST axis is strided but another axis is unit-stride (x.T, swapaxes)
    would be read one element per cache line by the generic kernel; the tiled transposing copy
    runs ~3x faster than that, and the streaming kernel then takes the contiguous copy."""
    st = x._strides
    if len(st) < 2 or st[-1] == 1 or st[-1] == 0:   # the common case first: this runs on every binary call
        return x
    if x.size >= (1 << 16) and x._expr is None and 1 in st[-3:-1] and x.shape[-1] >= 32:
        return _ccopy(x)
    return x


def _check_int_power(b):
    if isinstance(b, DeviceArray):
        if b.dtype.kind == "i" and py_bool(any(less(b, 0)).item()):
            raise ValueError("Integers to negative integer powers are not allowed.")
    elif isinstance(b, (int, np.integer)) and b < 0:
        raise ValueError("Integers to negative integer powers are not allowed.")


def _unary(ufunc, code, x):
    if not isinstance(x, DeviceArray):
        x = asarray(x)
    key = (ufunc, x.dtype)
    loop = _RESOLVE_CACHE.get(key)
    if loop is None:
        loop = ufunc.resolve_dtypes((x.dtype, None))
        for dt in loop:
            dtype_code(dt)
        _RESOLVE_CACHE[key] = loop
    # (sin(bool array): NumPy answers in float16 — computed in float32, rounded once, as NumPy's own half loops do: csrc/narrow.hip)
    if _LAZY and code != _capi.U_INVERT and x.size > 0 and x.size >= _LAZY_MIN and x._code < _NARROW_MIN:
        pcdt = _FLOAT_DT.get(loop[0])
        if pcdt is not None and (loop[1] == loop[0] or loop[1] == np.bool_) and x.dtype.kind in "fb" or \
                (pcdt is not None and x.dtype.kind == "i" and pcdt == _capi.F64):
            res = _lazy_node(_lz.UNARY, code, (x,), pcdt, x.shape, loop[1])
            if res is not None:
                return res
    res = DeviceArray._new(x.shape, loop[1])
    _lib().unary(code, x.desc(), res.desc())
    return res


def _extent(a: "DeviceArray"):
    """[lo, hi] element range of the block that the view touches."""
    lo = hi = a._offset
    for n, st in zip(a.shape, a._strides):
        if n > 1:
            if st > 0:
                hi += (n - 1) * st
            else:
                lo += (n - 1) * st
    return lo, hi


def _unalias(x, out: "DeviceArray"):
    """NumPy's ufunc machinery gives in-place calls whose operand OVERLAPS the destination the result of
    operating on a copy (`a[1:] += a[:-1]`, `a += a.T`); a kernel that reads and writes the same
    block concurrently would not. The identical view (`a += a`, `a *= a`) is safe element by element."""
    if not isinstance(x, DeviceArray) or x._buf is None or x._buf is not out._buf or x.size == 0:
        return x
    if x._offset == out._offset and x.shape == out.shape and x._strides == out._strides:
        return x
    (xl, xh), (ol, oh) = _extent(x), _extent(out)
    if xh < ol or oh < xl:
        return x
    return _ccopy(x)


def _py_int_for(dt, v):
    """A Python int on its way into an array of dtype `dt` (NEP 50): it must FIT an integer type (OverflowError otherwise); beyond
    64 bits it reaches a float array as the float it rounds to and a bool array as True — never as wrapped bits."""
    if dt.kind in "iu":
        info = np.iinfo(dt)
        if not (info.min <= v <= info.max):
            raise OverflowError(f"Python integer {v} out of bounds for {dt}")
    elif not (_INT64_MIN <= v <= _INT64_MAX):
        return py_bool(v) if dt.kind == "b" else float(v)
    return v


def _py_int_wrapped(dt, v):
    """.. and where NumPy CASTS a Python int instead (np.where, ufunc.at): modulo 2**bits into an integer type."""
    if dt.kind in "iu":
        info = np.iinfo(dt)
        if not (_INT64_MIN <= v < (1 << 64)):       # (NumPy converts to a C integer first: beyond 64 bits even a cast raises)
            raise OverflowError("Python int too large to convert to C long")
        if not (info.min <= v <= info.max):
            v &= (1 << info.bits) - 1
            if dt.kind == "i" and v > info.max:
                v -= 1 << info.bits
        return v
    return _py_int_for(dt, v)


def _copy_into(dst: DeviceArray, src, shape=None):
    """dst[...] = src with broadcasting and dtype conversion (unary COPY kernel)."""
    if type(src) is int:
        src = _py_int_for(dst.dtype, src)
    _before_write(dst)
    if dst._code >= _NARROW_MIN or (isinstance(src, DeviceArray) and src._code >= _NARROW_MIN) or (isinstance(src, np.generic) and src.dtype in _narrow._WIDE):
        if not isinstance(src, DeviceArray):     # a scalar: through a 0-d array of its own (or the default) type
            h = np.asarray(src)
            src = DeviceArray.from_numpy(h if h.dtype.kind != "O" else np.asarray(src, dtype=np.float64))
        src = _unalias(src, dst)
        _lib().convert(src.desc(dst.shape), dst.desc())
        return
    if isinstance(src, DeviceArray):
        src = _unalias(src, dst)
        sd = src.desc(dst.shape)
    else:
        src = _operand(src)
        if isinstance(src, DeviceArray):
            sd = src.desc(dst.shape)
        elif isinstance(src, np.generic):
            sd = _scalar_desc(src.item(), dtype_code(src.dtype))
        else:
            sd = _scalar_desc(src, _capi.F64 if isinstance(src, float) else _capi.I64)
    _lib().unary(_capi.U_COPY, sd, dst.desc())


def _fill(dst: DeviceArray, value):
    _before_write(dst)
    if type(value) is int:
        value = _py_int_for(dst.dtype, value)      # (a NumPy scalar converts instead)
    if isinstance(value, DeviceArray) or dst._code >= _NARROW_MIN:
        _copy_into(dst, value)
        return
    if isinstance(value, np.generic):
        value = value.item()
    if isinstance(value, complex):
        raise TypeError("complex fill values are not supported")
    code = _capi.F64 if isinstance(value, float) else _capi.I64
    _lib().fill(dst.desc(), _scalar_desc(value, code))


# =============================================================================
# elementwise API (names follow numpy / the backend table)
# =============================================================================
_DEFAULT_KW = {"where": (True,), "subok": (True, False), "order": ("K", "C", "A"), "casting": ("same_kind",), "copy": (True,), "initial": (), "like": (),
               "device": (), "signature": (), "mean": (), "correction": (), "out": (), "dtype": (), "ndmin": (0,)}


def _defaults_only(name, kw):
    """Keywords of the NumPy signature this backend has no code for: accepted at their defaults (None, or the values above), refused
    otherwise — a keyword silently dropped would answer a different question than the one asked."""
    for k, v in kw.items():
        if k not in _DEFAULT_KW:
            raise TypeError(f"{name}() got an unexpected keyword argument '{k}'")
        if v is not None and v is not np._NoValue and not builtins_any(v is d or (type(v) is type(d) and v == d) for d in _DEFAULT_KW[k]):
            raise TypeError(f"{name}(): the keyword {k} is not supported by the MI355X backend at this value ({type(v).__name__})")


def _finish_out(res, out, name):
    """`out=` of a reduction: NumPy's shape rule and same-kind cast, then the result lands in the caller's array."""
    if out is None:
        return res
    if not isinstance(out, DeviceArray):
        raise TypeError("output must be an array")
    if out.shape != res.shape:
        raise ValueError(f"output parameter for reduction operation {name} has the wrong shape: {out.shape} instead of {res.shape}")
    if not np.can_cast(res.dtype, out.dtype, casting="same_kind"):
        raise TypeError(f"Cannot cast ufunc '{name}' output from {res.dtype!r} to {out.dtype!r} with casting rule 'same_kind'")
    _copy_into(out, res)
    return out


def _ufunc_kwargs(name, out, kw):
    """NumPy's ufunc keywords: `out` (an array or a 1-tuple) is honoured; the others are accepted at their defaults and refused
    otherwise — a keyword silently dropped would answer a different question than the one asked."""
    for k, v in kw.items():
        if k not in ("where", "dtype", "casting", "order", "subok", "signature"):
            raise TypeError(f"{name}() got an unexpected keyword argument '{k}'")
        if not (v is None or (k in ("where", "subok") and v is True) or (k == "casting" and isinstance(v, str) and v == "same_kind") or
                (k == "order" and isinstance(v, str) and v == "K")):
            raise TypeError(f"{name}(): the keyword {k} is not supported by the MI355X backend at this value ({type(v).__name__})")
    if isinstance(out, tuple):
        if len(out) != 1:
            raise ValueError("The 'out' tuple must have exactly one entry per ufunc output")
        out = out[0]
    if out is not None and not isinstance(out, DeviceArray):
        raise TypeError("return arrays must be of ArrayType")
    return out


def _mk_unary(ufunc, code):
    def f(x, out=None, **kw):
        out = _ufunc_kwargs(ufunc.__name__, out, kw)
        res = _unary(ufunc, code, x)
        if out is None:
            return res
        if res.shape != out.shape:
            raise ValueError(f"non-broadcastable output operand with shape {out.shape} doesn't match the broadcast shape {res.shape}")
        if not np.can_cast(res.dtype, out.dtype, casting="same_kind"):
            raise np._core._exceptions._UFuncOutputCastingError(ufunc, "same_kind", res.dtype, out.dtype, 0)
        _copy_into(out, res)
        return out
    f.__name__ = ufunc.__name__
    return f


def _mk_binary(ufunc, code):
    def f(a, b, out=None, **kw):
        if out is not None or kw:
            out = _ufunc_kwargs(ufunc.__name__, out, kw)
        return _binary(ufunc, code, a, b, out=out)
    f.__name__ = ufunc.__name__
    return f


absolute = _mk_unary(np.absolute, _capi.U_ABS)
negative = _mk_unary(np.negative, _capi.U_NEG)
sign = _mk_unary(np.sign, _capi.U_SIGN)
ceil = _mk_unary(np.ceil, _capi.U_CEIL)
floor = _mk_unary(np.floor, _capi.U_FLOOR)
sin = _mk_unary(np.sin, _capi.U_SIN)
cos = _mk_unary(np.cos, _capi.U_COS)
tan = _mk_unary(np.tan, _capi.U_TAN)
sinh = _mk_unary(np.sinh, _capi.U_SINH)
cosh = _mk_unary(np.cosh, _capi.U_COSH)
tanh = _mk_unary(np.tanh, _capi.U_TANH)
exp = _mk_unary(np.exp, _capi.U_EXP)
log = _mk_unary(np.log, _capi.U_LOG)
sqrt = _mk_unary(np.sqrt, _capi.U_SQRT)
logical_not = _mk_unary(np.logical_not, _capi.U_LOGICAL_NOT)
invert = _mk_unary(np.invert, _capi.U_INVERT)
isnan = _mk_unary(np.isnan, _capi.U_ISNAN)

add = _mk_binary(np.add, _capi.B_ADD)
subtract = _mk_binary(np.subtract, _capi.B_SUB)
multiply = _mk_binary(np.multiply, _capi.B_MUL)
true_divide = _mk_binary(np.true_divide, _capi.B_TRUE_DIV)
floor_divide = _mk_binary(np.floor_divide, _capi.B_FLOOR_DIV)
mod = _mk_binary(np.remainder, _capi.B_MOD)
power = _mk_binary(np.power, _capi.B_POW)
maximum = _mk_binary(np.maximum, _capi.B_MAXIMUM)
minimum = _mk_binary(np.minimum, _capi.B_MINIMUM)
equal = _mk_binary(np.equal, _capi.B_EQ)
not_equal = _mk_binary(np.not_equal, _capi.B_NE)
less = _mk_binary(np.less, _capi.B_LT)
less_equal = _mk_binary(np.less_equal, _capi.B_LE)
greater = _mk_binary(np.greater, _capi.B_GT)
greater_equal = _mk_binary(np.greater_equal, _capi.B_GE)
logical_and = _mk_binary(np.logical_and, _capi.B_LAND)
logical_or = _mk_binary(np.logical_or, _capi.B_LOR)
logical_xor = _mk_binary(np.logical_xor, _capi.B_LXOR)


def where(condition, x=None, y=None):
    if x is None and y is None:
        return nonzero(condition)   # np.where(cond) is np.nonzero(cond)
    if x is None or y is None:
        raise ValueError("either both or neither of x and y should be given")
    c, a, b = _operand(condition), _operand(x), _operand(y)
    arrs = [v for v in (c, a, b) if isinstance(v, DeviceArray)]
    if not arrs:
        c = asarray(c)
        arrs = [c]
    # result dtype: np.result_type of the two branches (weak scalars included)
    ka, kb = _kind_key(a), _kind_key(b)
    key = ("where", ka, kb)
    odt = _RESOLVE_CACHE.get(key)
    if odt is None:
        def rt(k):
            return k if isinstance(k, np.dtype) else (0 if k is int else 0.0)
        odt = np.result_type(rt(ka), rt(kb))
        dtype_code(odt)
        _RESOLVE_CACHE[key] = odt
    shape = arrs[0].shape
    for v in arrs[1:]:
        shape = _broadcast_shapes(shape, v.shape)
    if _LAZY and _prod(shape) > 0 and _prod(shape) >= _LAZY_MIN and not builtins_any(v._code >= _NARROW_MIN for v in arrs):
        pcdt = _FLOAT_DT.get(odt)
        if pcdt is not None:
            res = _lazy_node(_lz.WHERE, 0, (c, a, b), pcdt, shape, odt)
            if res is not None:
                return res
    res = DeviceArray._new(shape, odt)
    dc = _operand_desc(c, shape, _capi.I64)
    # (np.where is not a ufunc: a Python int beyond the result type is CAST, wrapping, where a ufunc would raise OverflowError)
    if type(a) is int:
        a = _py_int_wrapped(odt, a)
    if type(b) is int:
        b = _py_int_wrapped(odt, b)
    da = _operand_desc(a, shape, 0 if isinstance(a, DeviceArray) else _scalar_code(a, odt))
    db = _operand_desc(b, shape, 0 if isinstance(b, DeviceArray) else _scalar_code(b, odt))
    _lib().where(dc, da, db, res.desc())
    return res


def clip(a, a_min=None, a_max=None, **kw):
    if "min" in kw:
        a_min = kw.pop("min")
    if "max" in kw:
        a_max = kw.pop("max")
    out = kw.pop("out", None)
    _defaults_only("clip", kw)
    res = a if isinstance(a, DeviceArray) else asarray(a)
    touched = False
    if a_min is not None:
        res = maximum(res, a_min)
        touched = True
    if a_max is not None:
        res = minimum(res, a_max)
        touched = True
    if out is not None:
        return _finish_out(res, out, "clip")
    return res if touched else _ccopy(res)


def astype(a, dtype, copy=True, **kw):
    if kw.get("casting") == "unsafe":      # (astype's own default)
        kw = {k: v for k, v in kw.items() if k != "casting"}
    _defaults_only("astype", kw)
    a = asarray(a)
    dtype = np.dtype(dtype)
    if dtype == a.dtype and not copy:
        return a
    res = DeviceArray.empty(a.shape, dtype)
    if a._code >= _NARROW_MIN or res._code >= _NARROW_MIN:
        _lib().convert(a.desc(), res.desc())
    else:
        _lib().unary(_capi.U_COPY, a.desc(), res.desc())
    return res


def _convert(a: "DeviceArray", dtype) -> "DeviceArray":
    """A new C-contiguous array of `dtype` with a's values converted (any pair of the 12 dtypes: mdhip_convert)."""
    res = DeviceArray.empty(a.shape, dtype)
    _lib().convert(a.desc(), res.desc())
    return res


def _ccopy(a):
    """A C-contiguous copy (ndarray.copy()'s default layout; what every internal "make it dense" wants)."""
    a = asarray(a)
    res = DeviceArray.empty(a.shape, a.dtype)
    if a._code >= _NARROW_MIN:
        _lib().convert(a.desc(), res.desc())
    else:
        _lib().unary(_capi.U_COPY, a.desc(), res.desc())
    return res


def _k_perm(a):
    """Axes from outermost to innermost in MEMORY — NumPy's order 'K' (nditer's axis ordering for one operand): an insertion sort
    by |stride| starting from C order in which a stride-0 (broadcast) axis compares as ambiguous and keeps its place."""
    st = a._strides
    perm = list(range(a.ndim - 1, -1, -1))          # fastest axis first, as nditer keeps them
    for i0 in range(1, a.ndim):
        pos, j0 = i0, perm[i0]
        for i1 in range(i0 - 1, -1, -1):
            s0, s1 = st[j0], st[perm[i1]]
            if s0 == 0 or s1 == 0:
                continue                            # ambiguous: look further, move only past a definite answer
            if abs(s1) <= abs(s0):
                break
            pos = i1
        if pos != i0:
            perm.insert(pos, perm.pop(i0))
    return tuple(reversed(perm))


def _resolve_order(a, order):
    """NumPy's memory-order argument for an existing array -> 'C', 'F' or 'K' ('A': Fortran only for an array that is
    F- and not C-contiguous; 'K' of a C- / F-contiguous array is C / F)."""
    if order not in ("C", "F", "A", "K"):
        raise ValueError("order must be one of 'C', 'F', 'A', or 'K'")
    if order in ("A", "K"):
        if a.is_c_contiguous or a.ndim < 2:
            return "C"
        if a.T.is_c_contiguous:
            return "F"
        return "C" if order == "A" else "K"
    return order


def copy(a, order="K", **kw):
    """np.copy: the values, laid out as `order` asks — 'K' (the default) keeps the source's axis order in memory, so a copy of x.T
    is a streaming copy into the same transposed layout, not a transposing one."""
    _defaults_only("copy", kw)
    a = asarray(a)
    order = _resolve_order(a, order)
    if order == "C":
        return _ccopy(a)
    # (the LAYOUT of a 'K' result follows PyArray_NewLikeArray: axes by descending |stride|, ties in axis order — a broadcast axis
    # goes innermost; the ORDER 'K' reads elements in is nditer's, _k_perm)
    perm = tuple(range(a.ndim - 1, -1, -1)) if order == "F" else tuple(sorted(range(a.ndim), key=lambda d: -abs(a._strides[d])))
    inv = [0] * a.ndim
    for i, p_ in enumerate(perm):
        inv[p_] = i
    return transpose(_ccopy(transpose(a, perm)), inv)


# =============================================================================
# layout (views wherever NumPy returns views)
# =============================================================================
def transpose(a, axes=None):
    a = asarray(a)
    nd = a.ndim
    if axes is None:
        perm = tuple(range(nd - 1, -1, -1))
    else:
        if isinstance(axes, DeviceArray):
            axes = axes.get().tolist()
        axes = [operator.index(x) for x in axes]
        if len(axes) != nd:
            raise ValueError("axes don't match array")
        perm = tuple(normalize_axis(x, nd) for x in axes)
        if len(set(perm)) != nd:
            raise ValueError("repeated axis in transpose")
    return a._view(a._offset, [a.shape[p] for p in perm], [a._strides[p] for p in perm])


def swapaxes(a, axis1, axis2):
    a = asarray(a)
    i, j = normalize_axis(axis1, a.ndim), normalize_axis(axis2, a.ndim)
    perm = list(range(a.ndim))
    perm[i], perm[j] = perm[j], perm[i]
    return transpose(a, perm)


def _reshape_view_strides(shape, strides, newshape):
    """Strides of a no-copy reshape, or None when the layout forbids it."""
    old = [(n, s) for n, s in zip(shape, strides) if n != 1]
    new_strides = [0] * len(newshape)
    oi = 0
    ni = 0
    nn = len(newshape)
    while ni < nn:
        if newshape[ni] == 1:
            new_strides[ni] = 1
            ni += 1
            continue
        if oi >= len(old):
            return None
        # grow a group of old dims and a group of new dims with equal products
        o0, n0 = oi, ni
        op, np_ = old[oi][0], newshape[ni]
        oi += 1
        ni += 1
        while op != np_:
            if op < np_:
                if oi >= len(old):
                    return None
                op *= old[oi][0]
                oi += 1
            else:
                while ni < nn and newshape[ni] == 1:
                    new_strides[ni] = 1
                    ni += 1
                if ni >= nn:
                    return None
                np_ *= newshape[ni]
                ni += 1
        # old[o0:oi] must be internally contiguous
        for k in range(o0, oi - 1):
            if old[k][1] != old[k + 1][1] * old[k + 1][0]:
                return None
        st = old[oi - 1][1]
        for k in range(ni - 1, n0 - 1, -1):
            if newshape[k] == 1:
                new_strides[k] = 1
                continue
            new_strides[k] = st
            st *= newshape[k]
    if oi != len(old):
        return None
    return new_strides


def reshape(a, shape=None, order="C", newshape=None, **kw):
    _defaults_only("reshape", kw)
    a = asarray(a)
    if shape is None:
        shape = newshape
    if order == "K":
        raise ValueError("order 'K' is not permitted for reshaping")
    if order is not None and _resolve_order(a, order) == "F":      # ('A': Fortran index order only for an F-contiguous array)
        return transpose(reshape(transpose(a), tuple(reversed(_normalize_shape(shape)))))
    shape = list(_normalize_shape(shape))
    size = a.size
    if shape.count(-1) > 1:
        raise ValueError("can only specify one unknown dimension")
    if -1 in shape:
        known = 1
        for s in shape:
            if s != -1:
                known *= s
        if known == 0 or size % known:
            raise ValueError(f"cannot reshape array of size {size} into shape {tuple(shape)}")
        shape[shape.index(-1)] = size // known
    if _prod(shape) != size:
        raise ValueError(f"cannot reshape array of size {size} into shape {tuple(shape)}")
    shape = tuple(shape)
    if len(shape) > MAX_NDIM:
        raise ValueError(f"maximum supported dimension for a DeviceArray is {MAX_NDIM}")
    if size == 0:
        return a._view(a._offset, shape, _c_strides(shape))
    st = _reshape_view_strides(a.shape, a._strides, shape)
    if st is None:
        a = _ccopy(a)
        st = _c_strides(shape)
    return a._view(a._offset, shape, st)


def ravel(a, order="C"):
    a = asarray(a)
    order = _resolve_order(a, order)
    if order == "F":
        return reshape(transpose(a), (-1,))
    if order == "K":      # memory order of a view that is neither C- nor F-contiguous
        return reshape(transpose(a, _k_perm(a)), (-1,))
    return reshape(a, (-1,))


def flatten(a, order="C"):
    a = asarray(a)
    r = ravel(a, order)
    return _ccopy(r) if r._buf is a._buf else r


def broadcast_to(a, shape, **kw):
    _defaults_only("broadcast_to", kw)
    a = asarray(a)
    shape = _normalize_shape(shape)
    if len(shape) < a.ndim:
        raise ValueError("input operand has more dimensions than allowed by the axis remapping")
    lead = len(shape) - a.ndim
    st = [0] * len(shape)
    for i, (n, s) in enumerate(zip(a.shape, a._strides)):
        tgt = shape[lead + i]
        if n == tgt:
            st[lead + i] = s
        elif n != 1:
            raise ValueError(f"operands could not be broadcast together with remapped shapes [original->remapped]: {a.shape} and requested shape {shape}")
    return a._view(a._offset, shape, st)


def expand_dims(a, axis):
    a = asarray(a)
    if isinstance(axis, DeviceArray):
        axis = axis.get().tolist()
    if not isinstance(axis, (tuple, list)):
        axis = (axis,)
    out_nd = a.ndim + len(axis)
    axes = normalize_axes(tuple(axis), out_nd)
    shape, strides = [], []
    it = iter(zip(a.shape, a._strides))
    for i in range(out_nd):
        if i in axes:
            shape.append(1)
            strides.append(0)
        else:
            n, s = next(it)
            shape.append(n)
            strides.append(s)
    return a._view(a._offset, shape, strides)


def squeeze(a, axis=None):
    a = asarray(a)
    if axis is None:
        keep = [i for i, n in enumerate(a.shape) if n != 1]
    else:
        if isinstance(axis, list):
            raise TypeError("'list' object cannot be interpreted as an integer")
        axes = normalize_axes(axis, a.ndim)
        for ax in axes:
            if a.shape[ax] != 1:
                raise ValueError("cannot select an axis to squeeze out which has size not equal to one")
        keep = [i for i in range(a.ndim) if i not in axes]
    return a._view(a._offset, [a.shape[i] for i in keep], [a._strides[i] for i in keep])


def flip(a, axis=None):
    a = asarray(a)
    axes = normalize_axes(axis, a.ndim)
    off = a._offset
    st = list(a._strides)
    for ax in axes:
        if a.shape[ax] > 0:
            off += (a.shape[ax] - 1) * st[ax]
        st[ax] = -st[ax]
    return a._view(off, a.shape, st)


def atleast_1d(a):
    a = asarray(a)
    return a if a.ndim >= 1 else reshape(a, (1,))


def atleast_2d(a):
    a = asarray(a)
    if a.ndim == 0:
        return reshape(a, (1, 1))
    if a.ndim == 1:
        return expand_dims(a, 0)
    return a


def atleast_3d(a):
    a = asarray(a)
    if a.ndim == 0:
        return reshape(a, (1, 1, 1))
    if a.ndim == 1:
        return expand_dims(a, (0, 2))
    if a.ndim == 2:
        return expand_dims(a, 2)
    return a


# =============================================================================
# reductions
# =============================================================================
def _reduce_axes(axis, a):
    """`axis` of a ufunc.reduce-style call: ints and tuples of ints (a list is a TypeError in NumPy); a 0-d operand takes 0 / -1."""
    if isinstance(axis, list):
        raise TypeError("'list' object cannot be interpreted as an integer")
    if a.ndim == 0 and isinstance(axis, (int, np.integer)) and not isinstance(axis, py_bool) and axis in (0, -1):
        return None
    return axis


def _reduce(code, a, axis, keepdims, out_dtype):
    a = asarray(a)
    axis = _reduce_axes(axis, a)
    axes = normalize_axes(axis, a.ndim)
    mask = 0
    for ax in axes:
        mask |= 1 << ax
    kshape = tuple(1 if (mask >> i) & 1 else n for i, n in enumerate(a.shape))
    if a._expr is not None:
        fused = _fused_reduce(code, a, mask, kshape, np.dtype(out_dtype))
        if fused is not None:
            if not keepdims:
                fshape = tuple(n for i, n in enumerate(a.shape) if not (mask >> i) & 1)
                fused = fused._view(fused._offset, fshape, _c_strides(fshape))
            return fused
    if a.size >= _RESHAPE_REDUCE_MIN and a.ndim >= 2:
        staged = _staged_reduce(code, a, axes, mask, keepdims, kshape, out_dtype)
        if staged is not None:
            return staged
    res = DeviceArray.empty(kshape, out_dtype)
    _lib().reduce(code, a.desc(), res.desc(), mask)
    if not keepdims:
        fshape = tuple(n for i, n in enumerate(a.shape) if not (mask >> i) & 1)
        res = res._view(res._offset, fshape, _c_strides(fshape))
    return res


_RESHAPE_REDUCE_MIN = 1 << 18


def _dense_flat_view(a):
    """1-D view over the block of an array whose axes are a permutation of a dense layout
    (x.T, swapaxes, ...): an order-free reduction may walk memory linearly instead."""
    dims = sorted((st, n) for st, n in zip(a._strides, a.shape) if n != 1)
    acc = 1
    for st, n in dims:
        if st != acc:
            return None
        acc *= n
    return a._view(a._offset, (a.size,), (1,))


def _staged_reduce(code, a, axes, mask, keepdims, kshape, out_dtype):
    """Large reductions the single-pass kernels walk badly, re-expressed through the fast ones
    (sum/prod/max/min/any/all are associative and commutative; float sums already differ
    from NumPy's pairwise order by rounding only):
      * every axis of a permuted-but-dense view  -> the same reduction over linear memory;
      * several separated groups of axes (e.g. (0, 2), the batch-norm (0, 2, 3)) -> one group
        per pass, innermost first, so every pass has ONE reduced extent (row or column walk)."""
    nd = a.ndim
    if mask == (1 << nd) - 1:
        if a.is_c_contiguous:
            return None
        flat = _dense_flat_view(a)
        if flat is None:
            return None
        r = _reduce(code, flat, None, False, out_dtype)
        shape = kshape if keepdims else ()
        return r._view(r._offset, shape, _c_strides(shape))
    runs = []  # maximal runs of adjacent reduced axes (extent-1 axes do not separate them)
    for ax in sorted(axes):
        if runs and builtins_all(a.shape[k] == 1 or (mask >> k) & 1 for k in range(runs[-1][-1] + 1, ax)):
            runs[-1].append(ax)
        else:
            runs.append([ax])
    if len(runs) < 2:
        return None
    cur = a
    for run in reversed(runs):
        cur = _reduce(code, cur, tuple(run), True, out_dtype)
    if not keepdims:
        fshape = tuple(n for i, n in enumerate(a.shape) if not (mask >> i) & 1)
        cur = cur._view(cur._offset, fshape, _c_strides(fshape))
    return cur


def _fused_reduce(code, a, mask, kshape, out_dtype):
    """reduce(pending expression) in one pass: full reductions and the axis-0
    reduce-to-shape of a 2-D expression; None -> caller materialises and reduces."""
    e = a._expr
    if e.kind == _lz.GEMM:
        return None
    if code not in (_capi.R_SUM, _capi.R_PROD, _capi.R_MAX, _capi.R_MIN):
        return None
    if _FLOAT_DT.get(out_dtype) != e.cdt or a.dtype != out_dtype or a.size == 0:
        return None
    nd = a.ndim
    full = mask == (1 << nd) - 1
    if full and code == _capi.R_SUM and e.kind == _lz.WHERE:
        res = _gemm_epilogue_sum(e, a, kshape, out_dtype)
        if res is not None:
            return res
    cols = nd == 2 and mask == 1 and a.shape[0] > 1 and a.shape[1] > 1 and a.shape[1] % 4 == 0
    if cols and not full and a.shape[1] < 512 and a.shape[0] >= 65536:
        # a TALL expression with few columns: the fused column kernels find their parallelism across columns (4,000,000 x 16: 6.2 ms
        # fused against 0.4 ms; 1,000,000 x 268: 1.7 against 1.0) — one fused evaluation, then the eager column reduction
        return None
    if not (full or cols):
        return None
    if cols and not full and a.size >= _DEFER_COLS_MIN and a.shape[1] % 1024 == 0 and a.shape[0] >= 512:
        # reduce-to-shape of an expression that is usually ALSO needed in memory (g * mask feeds the weight-gradient
        # GEMM and, column-summed, the bias gradient): owe the reduction until the expression is materialised, so that
        # both cost one pass (mdhip_vm_eval_reduce_cols); if the result is wanted first, it is computed alone
        res = DeviceArray.empty(kshape, out_dtype)
        task = _ColsTask(a, code, res)
        res._buf.task = task
        ts = a._tasks
        if ts is None:
            ts = a._tasks = []
        ts.append(weakref.ref(task))
        FUSION_STATS["deferred_cols"] += 1
        return res
    res = DeviceArray.empty(kshape, out_dtype)
    if not _reduce_pending(e, a.shape, code, mask, res):
        return None
    return res


_DEFER_COLS_MIN = 1 << 18


def _gemm_epilogue_sum(e, a, kshape, out_dtype):
    """sum(where(P + bias > 0, P + bias, 0)) with P a deferred matrix product X @ W and bias a row vector — the forward of
    BASELINE's MLP config as the untouched tape issues it (matmul, add, greater, where, sum). Runs as ONE GEMM whose
    epilogue adds the bias, accumulates the relu sum and writes the mask (mdhip_matmul_bias_relu_sum): the 128 MiB
    pre-activation never exists; the pending `greater` array — what the backward pass reads — becomes that mask.
    None: the expression is something else, or the shape is not covered (the caller runs the general path)."""
    c, t, f = e.args
    if f.kind != _lz.CONST or f.args != 0.0 or c.kind != _lz.BINARY or c.code != _capi.B_GT:
        return None
    if c.args[0] is not t or c.args[1].kind != _lz.CONST or c.args[1].args != 0.0:
        return None
    if t.kind != _lz.BINARY or t.code != _capi.B_ADD:
        return None
    p, q = t.args
    if p.kind != _lz.LEAF or q.kind != _lz.LEAF:
        return None
    if p.args._expr is None or p.args._expr.kind != _lz.GEMM:
        p, q = q, p
    prod, bias = p.args, q.args
    if prod._expr is None or prod._expr.kind != _lz.GEMM or bias._expr is not None:
        return None
    if prod.shape != a.shape or bias.dtype != np.float32 or out_dtype != np.float32:
        return None
    M, N = prod.shape
    if bias.shape == (1, N):
        bias = bias._view(bias._offset, (N,), bias._strides[1:])
    if bias.shape != (N,) or bias._strides != (1,):
        return None
    x, w = prod._expr.args
    if x._strides[-1] != 1 or w._strides[-1] != 1:
        return None
    mask = DeviceArray.empty((M, N), np.bool_)
    res = DeviceArray.empty(kshape, out_dtype)
    try:
        _lib().matmul_bias_relu_sum(x.desc(), w.desc(), bias.desc(), mask.desc(), res.desc())
    except ValueError:
        return None
    FUSION_STATS["gemm_epilogue"] += 1
    owner = c.owner() if c.owner is not None else None
    if owner is not None and owner._expr is c and owner.dtype == np.bool_ and owner.shape == (M, N):
        owner._buf, owner._expr, owner._cdesc = mask._buf, None, None   # `z > 0` is now in memory
    return res


def _reduce_pending(e, shape, code, mask, res) -> py_bool:
    prog, keep = _lz.build_program(e, shape)
    nd = len(shape)
    shape_like = ArrayDesc()
    shape_like.dtype = e.cdt
    shape_like.ndim = nd
    if nd:
        shape_like.shape[:nd] = shape
    try:
        _lib().vm_reduce(prog, code, shape_like, res.desc(), mask)
    except ValueError:
        return False
    FUSION_STATS["vm_reduce"] += 1
    del keep
    return True


class _ColsTask:
    """A column reduction (axis 0) of a pending 2-D expression whose result block exists but is not filled yet.
    Owned by the result's block (`_Buffer.task`) and referring back to it weakly (no cycle: dropping the result
    drops the task and releases the expression); the source array keeps a weak reference too. Runs (a) fused
    into the source's materialisation, or (b) alone, the moment anything needs the result's bytes."""

    __slots__ = ("src", "code", "_buf", "_kshape", "_dtype", "done", "__weakref__")

    def __init__(self, src, code, res):
        self.src, self.code, self.done = src, code, False
        self._buf, self._kshape, self._dtype = weakref.ref(res._buf), res.shape, res.dtype

    @property
    def res(self):
        return DeviceArray(self._buf(), 0, self._kshape, _c_strides(self._kshape), self._dtype)

    def claim(self):
        """Mark as being carried out by the caller (who then fills `res`)."""
        self.done = True
        b = self._buf()
        if b is not None:
            b.task = None
        self.src = None

    def unclaim(self, src):
        """The pass that claimed the task failed before filling `res`: owed again."""
        self.done = False
        self.src = src
        b = self._buf()
        if b is not None:
            b.task = self

    def run(self):
        if self.done:
            return
        a, res = self.src, self.res
        self.claim()
        e = a._expr
        if e is None:  # (materialised by a path that did not see the task)
            _lib().reduce(self.code, a.desc(), res.desc(), 1)
        elif not _reduce_pending(e, a.shape, self.code, 1, res):
            a._materialize()
            _lib().reduce(self.code, a.desc(), res.desc(), 1)


def _sum_dtype(a_dtype, dtype):
    if dtype is not None:
        return np.dtype(dtype)
    if a_dtype.kind in "bi":
        return np.dtype(np.int64)
    if a_dtype.kind == "u":
        return np.dtype(np.uint64)      # (NumPy: unsigned integers accumulate in the unsigned platform integer)
    return a_dtype


def sum(a, axis=None, dtype=None, out=None, keepdims=False, **kw):
    _defaults_only("sum", kw)
    a = asarray(a)
    return _finish_out(_reduce(_capi.R_SUM, a, axis, keepdims, _sum_dtype(a.dtype, dtype)), out, "add")


def prod(a, axis=None, dtype=None, out=None, keepdims=False, **kw):
    _defaults_only("prod", kw)
    a = asarray(a)
    return _finish_out(_reduce(_capi.R_PROD, a, axis, keepdims, _sum_dtype(a.dtype, dtype)), out, "multiply")


def max(a, axis=None, out=None, keepdims=False, **kw):
    _defaults_only("max", kw)
    a = asarray(a)
    return _finish_out(_reduce(_capi.R_MAX, a, axis, keepdims, a.dtype), out, "maximum")


def min(a, axis=None, out=None, keepdims=False, **kw):
    _defaults_only("min", kw)
    a = asarray(a)
    return _finish_out(_reduce(_capi.R_MIN, a, axis, keepdims, a.dtype), out, "minimum")


def any(a, axis=None, out=None, keepdims=False, **kw):
    _defaults_only("any", kw)
    return _finish_out(_reduce(_capi.R_ANY, a, axis, keepdims, np.dtype(np.bool_)), out, "logical_or")


def all(a, axis=None, out=None, keepdims=False, **kw):
    _defaults_only("all", kw)
    return _finish_out(_reduce(_capi.R_ALL, a, axis, keepdims, np.dtype(np.bool_)), out, "logical_and")


def _arg_reduce(code, a, axis, keepdims):
    a = asarray(a)
    if axis is not None and not isinstance(axis, (int, np.integer)):
        raise TypeError(f"'{type(axis).__name__}' object cannot be interpreted as an integer")
    if axis is None:
        flat = ravel(a)
        res = _reduce(code, flat, 0, False, np.dtype(np.int64))
        if keepdims:
            res = reshape(res, (1,) * a.ndim)
        return res
    ax = normalize_axis(int(axis), a.ndim) if a.ndim else int(axis)
    if a.ndim >= 2 and a._expr is None and a.shape[ax] >= 16384 and a.size // a.shape[ax] <= 64 and a._strides[ax] != 1:
        # a handful of LONG strided lines (argmax down the two columns of a 2,000,000 x 2 array: 4.3 ms on the column-strips kernel, which
        # finds its parallelism across columns): gather the lines into rows first — 16 MB copied in 0.05 ms — and search each row
        perm = [d for d in range(a.ndim) if d != ax] + [ax]
        res = _reduce(code, _ccopy(transpose(a, perm)), a.ndim - 1, False, np.dtype(np.int64))
        return expand_dims(res, ax) if keepdims else res
    return _reduce(code, a, int(axis), keepdims, np.dtype(np.int64))


def argmax(a, axis=None, out=None, keepdims=False, **kw):
    _defaults_only("argmax", kw)
    return _finish_out(_arg_reduce(_capi.R_ARGMAX, a, axis, keepdims), out, "argmax")


def argmin(a, axis=None, out=None, keepdims=False, **kw):
    _defaults_only("argmin", kw)
    return _finish_out(_arg_reduce(_capi.R_ARGMIN, a, axis, keepdims), out, "argmin")


def _count(a, axis):
    if isinstance(axis, list):
        raise TypeError("'list' object cannot be interpreted as an integer")
    axes = normalize_axes(axis, a.ndim)
    n = 1
    for ax in axes:
        n *= a.shape[ax]
    return n


def mean(a, axis=None, dtype=None, out=None, keepdims=False, **kw):
    _defaults_only("mean", kw)
    # numpy/_core/_methods.py:_mean — sum in the float dtype, then true_divide by the count
    a = asarray(a)
    n = _count(a, axis)
    if dtype is None and a.dtype.kind in "bi":
        dtype = np.dtype(np.float64)
    s = sum(a, axis=axis, dtype=dtype, keepdims=keepdims)
    if s.dtype.kind in "iub":
        # an integer dtype= : NumPy divides with casting='unsafe' back into the integer sum (the quotient truncates toward zero)
        with np.errstate(all="ignore"):
            return _finish_out(astype(_binary(np.true_divide, _capi.B_TRUE_DIV, s, n), s.dtype), out, "mean")
    return _finish_out(_binary(np.true_divide, _capi.B_TRUE_DIV, s, n, out=s), out, "mean")


def _std_fused(a, axis, dtype, ddof, keepdims, n):
    """One-pass (rows) / two-pass (columns) kernel for the forms mdhip_var covers: a concrete float32 / float64 C-contiguous array
    reduced over its last axis, or a 2-D one over its first; anything else -> None and `std` composes NumPy's five steps."""
    if a._code not in _FLOAT_CODES or a._expr is not None or (dtype is not None and np.dtype(dtype) != a.dtype) or a.ndim == 0:
        return None
    if isinstance(axis, (tuple, list)):
        if len(axis) != 1:
            return None
        axis = axis[0]
    if axis is None:
        if a.ndim != 1:
            return None
        axis = 0
    if not isinstance(axis, (int, np.integer)) or isinstance(axis, py_bool) or not isinstance(ddof, (int, np.integer)) or isinstance(ddof, py_bool):
        return None
    ax = int(axis)
    if ax < -a.ndim or ax >= a.ndim:
        return None          # (the composed path raises NumPy's AxisError)
    ax %= a.ndim
    if n - int(ddof) <= 0 or n < 2 or not a.is_c_contiguous or a.size == 0:
        return None
    if not (ax == a.ndim - 1 or (ax == 0 and a.ndim == 2)):
        return None
    kshape = a.shape[:ax] + (1,) + a.shape[ax + 1:]
    res = DeviceArray._new(kshape, a.dtype)
    try:
        _lib().var(a.desc(), res.desc(), ax, int(ddof), 1)
    except ValueError:       # a form the kernel leaves to the composition (alignment, short / narrow shapes)
        return None
    return res if keepdims else reshape(res, a.shape[:ax] + a.shape[ax + 1:])


def std(a, axis=None, dtype=None, out=None, ddof=0, keepdims=False, **kw):
    _defaults_only("std", kw)
    # numpy/_core/_methods.py:_var/_std — mean, centred squares, mean, sqrt
    a = asarray(a)
    n = _count(a, axis)
    if dtype is None and a.dtype.kind in "bi":
        dtype = np.dtype(np.float64)
    fused = _std_fused(a, axis, dtype, ddof, keepdims, n)
    if fused is not None:
        return _finish_out(fused, out, "std")
    arrmean = sum(a, axis=axis, dtype=dtype, keepdims=True)
    arrmean = _binary(np.true_divide, _capi.B_TRUE_DIV, arrmean, n, out=arrmean)
    x = subtract(a, arrmean)
    x = multiply(x, x)
    ret = sum(x, axis=axis, dtype=dtype, keepdims=keepdims)
    rcount = builtins_max(n - ddof, 0)
    ret = _binary(np.true_divide, _capi.B_TRUE_DIV, ret, rcount, out=ret)
    return _finish_out(sqrt(ret), out, "std")


# =============================================================================
# matmul family
# =============================================================================
def _as3d(x: DeviceArray, batch_shape: tuple):
    """View x (..., r, c) as (B, r, c) with B = prod(batch_shape) (broadcast)."""
    r, c = x.shape[-2], x.shape[-1]
    xb = x.shape[:-2]
    if not batch_shape:
        return x._view(x._offset, (1, r, c), (0, x._strides[-2], x._strides[-1]))
    full = broadcast_to(x, batch_shape + (r, c)) if xb != batch_shape else x
    B = _prod(batch_shape)
    # collapse batch dims into one if strides allow, else copy
    st = _reshape_view_strides(full.shape[:-2], full._strides[:-2], (B,)) if B > 1 else [0]
    if st is None:
        if _prod(xb) == 1 or builtins_all(s == 0 for s in full._strides[:-2]):
            st = [0]
        else:
            full = _ccopy(full)
            st = [r * c]
    return full._view(full._offset, (B, r, c), (st[0], full._strides[-2], full._strides[-1]))


def matmul(a, b, out=None, **kw):
    """np.matmul (numpy.py:84). `out`: like NumPy's — a C-contiguous array of the result's shape and dtype that
    receives the product (dp.GradSync points it at a row panel of the all-reduce bucket)."""
    _defaults_only("matmul", kw)
    a, b = asarray(a), asarray(b)
    if a.ndim == 0 or b.ndim == 0:
        raise ValueError("matmul: Input operand does not have enough dimensions (has 0, gufunc core with signature (n?,k),(k,m?)->(n?,m?) requires 1)")
    odt = np.result_type(a.dtype, b.dtype)
    if odt == np.bool_:
        # NumPy's boolean loop: "any k with a[i, k] and b[k, j]" — the integer product of the 0 / 1 operands is the count of such k
        res = greater(matmul(astype(a, np.int32), astype(b, np.int32)), 0)
        return _finish_out(res, out, "matmul")
    dtype_code(odt)
    if a.dtype != odt:
        a = astype(a, odt)
    if b.dtype != odt:
        b = astype(b, odt)
    a_vec, b_vec = a.ndim == 1, b.ndim == 1
    if a_vec:
        a = expand_dims(a, 0)
    if b_vec:
        b = expand_dims(b, 1)
    if a.shape[-1] != b.shape[-2]:
        raise ValueError(
            f"matmul: Input operand 1 has a mismatch in its core dimension 0, with gufunc signature (n?,k),(k,m?)->(n?,m?) (size {b.shape[-2]} is different from {a.shape[-1]})")
    batch = _broadcast_shapes(a.shape[:-2], b.shape[:-2])
    M, N = a.shape[-2], b.shape[-1]
    if _LAZY and out is None and not batch and not a_vec and not b_vec and odt == np.float32 and a.shape[-1] > 0 and M * N > 0:
        # deferred: the product may end up in the epilogue-fused form (_fused_reduce), otherwise it runs
        # unchanged the moment anything needs its bytes
        a.materialize()
        b.materialize()
        return DeviceArray._pending_gemm(a, b, (M, N), odt, _capi.F32)
    if out is not None:
        if not isinstance(out, DeviceArray) or a_vec or b_vec or out.shape != batch + (M, N) or out.dtype != odt or not out.is_c_contiguous:
            raise ValueError("matmul: out must be a C-contiguous DeviceArray with the shape and dtype of the result")
        _before_write(out)
        out = _unalias_out(out, a, b)
        res = out
    else:
        res = DeviceArray.empty(batch + (M, N), odt)
    B = _prod(batch)
    a3 = _as3d(a, batch)
    b3 = _as3d(b, batch)
    c3 = res._view(res._offset, (B, M, N), (M * N, N, 1))
    if a.shape[-1] == 0:
        _fill(res, 0)
    else:
        _lib().matmul(a3.desc(), b3.desc(), c3.desc())
    if a_vec and b_vec:
        return res._view(res._offset, batch, _c_strides(batch))
    if a_vec:
        shp = batch + (N,)
        return res._view(res._offset, shp, _c_strides(shp))
    if b_vec:
        shp = batch + (M,)
        return res._view(res._offset, shp, _c_strides(shp))
    return res


def _gemm_into(a, b, res):
    """res[...] = a @ b for 2-D operands (the launch of a deferred product)."""
    M, N = res.shape
    a3 = a._view(a._offset, (1,) + a.shape, (0,) + a._strides)
    b3 = b._view(b._offset, (1,) + b.shape, (0,) + b._strides)
    c3 = res._view(res._offset, (1, M, N), (M * N, N, 1))
    _lib().matmul(a3.desc(), b3.desc(), c3.desc())


def _unalias_out(out, *operands):
    """matmul reads its operands while it writes: an `out` that shares a block with one of them is refused."""
    for x in operands:
        if x._buf is not None and x._buf is out._buf:
            raise ValueError("matmul: out must not overlap an operand")
    return out


def dot(a, b, **kw):
    _defaults_only("dot", kw)
    a_s = not isinstance(a, DeviceArray) and np.ndim(a) == 0
    b_s = not isinstance(b, DeviceArray) and np.ndim(b) == 0
    if a_s or b_s:
        return multiply(a, b)
    a, b = asarray(a), asarray(b)
    if a.ndim == 0 or b.ndim == 0:
        return multiply(a, b)
    if a.ndim <= 2 and b.ndim <= 2:
        if a.shape[-1] != (b.shape[0] if b.ndim == 1 else b.shape[-2]):
            raise ValueError(f"shapes {a.shape} and {b.shape} not aligned: {a.shape[-1]} (dim {a.ndim - 1}) != {b.shape[0] if b.ndim == 1 else b.shape[-2]} (dim {0 if b.ndim == 1 else b.ndim - 2})")
        return matmul(a, b)
    if b.ndim == 1:
        return tensordot(a, b, axes=([a.ndim - 1], [0]))
    return tensordot(a, b, axes=([a.ndim - 1], [b.ndim - 2]))


def tensordot(a, b, axes=2):
    # numpy/_core/numeric.py:tensordot restated: move contracted axes together,
    # flatten to 2-D, one GEMM, reshape back.
    a, b = asarray(a), asarray(b)
    try:
        iter(axes)
    except TypeError:
        n = int(axes)
        axes_a = list(range(-n, 0))
        axes_b = list(range(0, n))
    else:
        axes_a, axes_b = axes
    if isinstance(axes_a, DeviceArray):
        axes_a = axes_a.get().tolist()
    if isinstance(axes_b, DeviceArray):
        axes_b = axes_b.get().tolist()
    try:
        axes_a = list(axes_a)
    except TypeError:
        axes_a = [axes_a]
    try:
        axes_b = list(axes_b)
    except TypeError:
        axes_b = [axes_b]
    if len(axes_a) != len(axes_b):
        raise ValueError("shape-mismatch for sum")
    axes_a = [normalize_axis(x, a.ndim) if -a.ndim <= x < a.ndim else x for x in axes_a]
    axes_b = [normalize_axis(x, b.ndim) if -b.ndim <= x < b.ndim else x for x in axes_b]
    for xa, xb in zip(axes_a, axes_b):
        if a.shape[xa] != b.shape[xb]:
            raise ValueError("shape-mismatch for sum")
    free_a = [k for k in range(a.ndim) if k not in axes_a]
    free_b = [k for k in range(b.ndim) if k not in axes_b]
    n2 = _prod([a.shape[k] for k in axes_a])
    at = reshape(transpose(a, free_a + axes_a), (_prod([a.shape[k] for k in free_a]), n2))
    bt = reshape(transpose(b, axes_b + free_b), (n2, _prod([b.shape[k] for k in free_b])))
    res = matmul(at, bt)
    return reshape(res, tuple(a.shape[k] for k in free_a) + tuple(b.shape[k] for k in free_b))


# =============================================================================
# creation
# =============================================================================
_DEFAULT_FLOAT = np.dtype(np.float64)


def zeros(shape, dtype=None, **kw):
    _defaults_only("zeros", kw)
    res = DeviceArray.empty(shape, dtype or _DEFAULT_FLOAT)
    _fill(res, 0)
    return res


def ones(shape, dtype=None, **kw):
    _defaults_only("ones", kw)
    res = DeviceArray.empty(shape, dtype or _DEFAULT_FLOAT)
    _fill(res, 1)
    return res


def full(shape, fill_value, dtype=None, **kw):
    _defaults_only("full", kw)
    if dtype is None:
        dtype = fill_value.dtype if isinstance(fill_value, (DeviceArray, np.generic)) else np.array(fill_value).dtype
    res = DeviceArray.empty(shape, dtype)
    _fill(res, fill_value)
    return res


def _like_dtype(a, dtype):
    if dtype is not None:
        return np.dtype(dtype)
    if isinstance(a, DeviceArray):
        return a.dtype
    return np.asarray(a).dtype


def _like_shape(a, shape):
    if shape is not None:
        return _normalize_shape(shape)
    return a.shape if isinstance(a, DeviceArray) else np.shape(a)


def zeros_like(a, dtype=None, shape=None, **kw):
    _defaults_only("zeros_like", kw)
    return zeros(_like_shape(a, shape), _like_dtype(a, dtype))


def ones_like(a, dtype=None, shape=None, **kw):
    _defaults_only("ones_like", kw)
    return ones(_like_shape(a, shape), _like_dtype(a, dtype))


def full_like(a, fill_value, dtype=None, shape=None, **kw):
    _defaults_only("full_like", kw)
    return full(_like_shape(a, shape), fill_value, _like_dtype(a, dtype))


def arange(*args, dtype=None, **kw):
    _defaults_only("arange", kw)
    args = [x.item() if isinstance(x, (DeviceArray, np.generic)) else x for x in args]
    if len(args) == 1:
        start, stop, step = 0, args[0], 1
    elif len(args) == 2:
        start, stop, step = args[0], args[1], 1
    elif len(args) == 3:
        start, stop, step = args
    else:
        raise TypeError("arange() requires 1-3 positional arguments")
    if step == 0:
        raise ZeroDivisionError("division by zero")
    if dtype is None:
        dtype = np.result_type(*[type(v)(0) if isinstance(v, (int, float)) else v for v in (start, stop, step)])
        if dtype.kind == "b":
            dtype = np.dtype(np.int64)
    dtype = np.dtype(dtype)
    n = int(math.ceil((stop - start) / step))
    if n < 0:
        n = 0
    res = DeviceArray.empty((n,), dtype)
    if n:
        _lib().arange(res.desc(), float(start), float(step))
    return res


def concatenate(arrays, axis=0, **kw):
    _defaults_only("concatenate", kw)
    arrays = [asarray(x) for x in arrays]
    if not arrays:
        raise ValueError("need at least one array to concatenate")
    if axis is None:
        arrays = [ravel(x) for x in arrays]
        axis = 0
    nd = arrays[0].ndim
    if nd == 0:
        raise ValueError("zero-dimensional arrays cannot be concatenated")
    ax = normalize_axis(axis, nd)
    odt = np.result_type(*[x.dtype for x in arrays])
    base = list(arrays[0].shape)
    total = 0
    for k, x in enumerate(arrays):
        if x.ndim != nd:
            raise ValueError(f"all the input array dimensions except for the concatenation axis must match exactly, but along dimension 0, the array at index 0 has {nd} dimension(s) and the array at index {k} has {x.ndim} dimension(s)")
        for d in range(nd):
            if d != ax and x.shape[d] != base[d]:
                raise ValueError(f"all the input array dimensions except for the concatenation axis must match exactly, but along dimension {d}, the array at index 0 has size {base[d]} and the array at index {k} has size {x.shape[d]}")
        total += x.shape[ax]
    base[ax] = total
    res = DeviceArray.empty(tuple(base), odt)
    pos = 0
    for x in arrays:
        n = x.shape[ax]
        if n:
            shp = list(res.shape)
            shp[ax] = n
            dst = res._view(res._offset + pos * res._strides[ax], shp, res._strides)
            _copy_into(dst, x)
        pos += n
    return res


def stack(arrays, axis=0, **kw):
    _defaults_only("stack", kw)
    arrays = [asarray(x) for x in arrays]
    if not arrays:
        raise ValueError("need at least one array to stack")
    shp = arrays[0].shape
    for x in arrays:
        if x.shape != shp:
            raise ValueError("all input arrays must have the same shape")
    ax = normalize_axis(axis, len(shp) + 1)
    return concatenate([expand_dims(x, ax) for x in arrays], axis=ax)


def tile(A, reps):
    A = asarray(A)
    if isinstance(reps, DeviceArray):
        reps = reps.get().tolist()
    try:
        reps = tuple(int(r) for r in reps)
    except TypeError:
        reps = (int(reps),)
    d = len(reps)
    if d < A.ndim:
        reps = (1,) * (A.ndim - d) + reps
    if A.ndim < len(reps):
        A = reshape(A, (1,) * (len(reps) - A.ndim) + A.shape)
    nd = A.ndim
    if 2 * nd > MAX_NDIM:
        # fall back to one axis at a time to stay within the descriptor rank
        res = A
        for ax, r in enumerate(reps):
            if r != 1:
                res = concatenate([res] * r, axis=ax) if r > 0 else res._view(res._offset, res.shape[:ax] + (0,) + res.shape[ax + 1:], res._strides)
        return _ccopy(res) if res is A else res
    # (r0, a0, r1, a1, ...) broadcast view -> one strided copy
    ishape, istr = [], []
    for r, n, s in zip(reps, A.shape, A._strides):
        ishape += [r, n]
        istr += [0, s]
    view = A._view(A._offset, ishape, istr)
    res = DeviceArray.empty(tuple(ishape), A.dtype)
    _lib().unary(_capi.U_COPY, view.desc(), res.desc())
    fshape = tuple(r * n for r, n in zip(reps, A.shape))
    return res._view(res._offset, fshape, _c_strides(fshape))


def repeat(a, repeats, axis=None):
    a = asarray(a)
    if axis is None:
        a = ravel(a)
        axis = 0
    ax = normalize_axis(axis, a.ndim)
    if isinstance(repeats, DeviceArray):
        repeats = repeats.get()
    if np.ndim(repeats) == 0:
        r = int(repeats)
        shp = a.shape[:ax + 1] + (r,) + a.shape[ax + 1:]
        st = a._strides[:ax + 1] + (0,) + a._strides[ax + 1:]
        if len(shp) > MAX_NDIM:
            raise ValueError("repeat: rank too large for the device descriptor")
        view = a._view(a._offset, shp, st)
        res = _ccopy(view)
        fshape = a.shape[:ax] + (a.shape[ax] * r,) + a.shape[ax + 1:]
        return res._view(res._offset, fshape, _c_strides(fshape))
    idx = np.repeat(np.arange(a.shape[ax]), np.asarray(repeats))
    key = (slice(None),) * ax + (asarray(idx),)
    return getitem(a, key)


def split(ary, indices_or_sections, axis=0):
    ary = asarray(ary)
    ax = normalize_axis(axis, ary.ndim)
    n = ary.shape[ax]
    if isinstance(indices_or_sections, DeviceArray):
        indices_or_sections = indices_or_sections.get().tolist()
    if isinstance(indices_or_sections, (int, np.integer)):
        k = int(indices_or_sections)
        if k <= 0 or n % k:
            raise ValueError("array split does not result in an equal division")
        cuts = [i * (n // k) for i in range(1, k)]
    else:
        cuts = [int(c) for c in indices_or_sections]
    out = []
    prev = 0
    for c in cuts + [n]:
        c = builtins_min(builtins_max(c if c >= 0 else c + n, 0), n)
        lo = builtins_min(prev, n)
        key = (slice(None),) * ax + (slice(lo, builtins_max(c, lo)),)
        out.append(getitem(ary, key))
        prev = c
    return out



# =============================================================================
# indexing (bit-exact data movement)
# =============================================================================
def flatnonzero(a) -> "DeviceArray":
    """Ascending flat positions of the non-zero elements (int64), computed on the device
    (count -> scan -> ordered compaction; one host sync for the data-dependent size)."""
    a = asarray(a)
    if not a.is_c_contiguous:
        a = _ccopy(a)
    cnt = C.c_int64()
    _lib().nonzero_count(a.desc(), C.byref(cnt))
    res = DeviceArray.empty((cnt.value,), np.int64)
    if cnt.value:
        _lib().nonzero_fill(a.desc(), cnt.value, res.ptr)
    return res


def _unravel(flat: "DeviceArray", shape) -> tuple:
    """flat int64 positions -> one index array per axis of `shape` (device integer arithmetic)."""
    out = []
    rem = flat
    for n in reversed(shape[1:]):
        out.append(mod(rem, n))
        rem = floor_divide(rem, n)
    out.append(rem)
    return tuple(reversed(out))


def nonzero(a) -> tuple:
    a = asarray(a)
    if a.ndim == 0:
        raise ValueError("Calling nonzero on 0d arrays is not allowed. Use np.atleast_1d(scalar).nonzero() instead.")
    return _unravel(flatnonzero(a), a.shape)


def _parse_key(a: DeviceArray, key):
    """-> (entries, has_advanced). entries: ('slice', slice) | ('new',) | ('adv', DeviceArray|int)"""
    if not isinstance(key, tuple):
        key = (key,)
    expanded = []
    for k in key:
        if isinstance(k, list):
            k = np.array(k)
        if isinstance(k, np.ndarray):
            if k.dtype == np.bool_:
                k = DeviceArray.from_numpy(k)
            elif k.dtype.kind in "iu":
                k = DeviceArray.from_numpy(k.astype(np.int64))
            else:
                raise IndexError("arrays used as indices must be of integer (or boolean) type")
        if isinstance(k, (py_bool, np.bool_)) or (isinstance(k, DeviceArray) and k.dtype == np.bool_ and k.ndim == 0):
            # a scalar boolean consumes no axis: it ADDS one, of length 1 (True) or 0 (False) — NumPy's 0-d boolean index
            expanded.append(("bool0", py_bool(k.item() if isinstance(k, DeviceArray) else k)))
        elif isinstance(k, DeviceArray) and k.dtype == np.bool_:
            expanded.append(("bool", k))
        else:
            expanded.append(k)
    n_consume = 0
    for k in expanded:
        if k is None or k is Ellipsis:
            continue
        if isinstance(k, tuple) and k[0] == "bool":
            n_consume += k[1].ndim
        elif isinstance(k, tuple) and k[0] == "bool0":
            pass
        else:
            n_consume += 1
    if builtins_sum(1 for k in expanded if k is Ellipsis) > 1:
        raise IndexError("an index can only have a single ellipsis ('...')")
    if n_consume > a.ndim:
        raise IndexError(f"too many indices for array: array is {a.ndim}-dimensional, but {n_consume} were indexed")
    entries = []
    has_adv = False
    saw_ellipsis = False
    for k in expanded:
        if k is Ellipsis:
            saw_ellipsis = True
            entries += [("slice", slice(None))] * (a.ndim - n_consume)
        elif k is None:
            entries.append(("new",))
        elif isinstance(k, slice):
            entries.append(("slice", k))
        elif isinstance(k, tuple) and k[0] == "bool0":
            entries.append(("new",) if k[1] else ("new0",))
        elif isinstance(k, tuple) and k[0] == "bool":
            m = k[1]
            ax0 = builtins_sum(1 for e in entries if e[0] not in ("new", "new0"))
            if m.shape != a.shape[ax0:ax0 + m.ndim]:
                raise IndexError(f"boolean index did not match indexed array along axis {ax0}; size of axis is {a.shape[ax0]} but size of corresponding boolean axis is {m.shape[0]}")
            for ix in nonzero(m):
                entries.append(("adv", ix))
            has_adv = True
        elif isinstance(k, DeviceArray):
            if k.dtype.kind not in "iu":
                raise IndexError("arrays used as indices must be of integer (or boolean) type")
            entries.append(("adv", k))
            has_adv = True
        elif isinstance(k, (int, np.integer)):
            entries.append(("int", int(k)))
        else:
            try:
                entries.append(("int", operator.index(k)))
            except TypeError:
                raise IndexError("only integers, slices (`:`), ellipsis (`...`), numpy.newaxis (`None`) and integer or boolean arrays are valid indices")
    if not saw_ellipsis:
        entries += [("slice", slice(None))] * (a.ndim - n_consume)
    if has_adv and builtins_any(isinstance(k, tuple) and k[0] == "bool0" for k in expanded):
        raise IndexError("a scalar boolean index together with index arrays is not supported on the device")
    return entries, has_adv


def _basic_view(a: DeviceArray, entries):
    off = a._offset
    shape, strides = [], []
    ax = 0
    for e in entries:
        kind = e[0]
        if kind == "new" or kind == "new0":
            shape.append(1 if kind == "new" else 0)
            strides.append(0)
            continue
        n, s = a.shape[ax], a._strides[ax]
        if kind == "int":
            i = e[1]
            if i < -n or i >= n:
                raise IndexError(f"index {i} is out of bounds for axis {ax} with size {n}")
            off += (i + n if i < 0 else i) * s
        else:
            start, stop, step = e[1].indices(n)
            cnt = len(range(start, stop, step))
            off += start * s if cnt else 0
            shape.append(cnt)
            strides.append(s * step)
        ax += 1
    return a._view(off, shape, strides)


def _build_plan(a: DeviceArray, entries):
    """Index plan for a key with integer-array entries (NumPy advanced indexing
    rules: ints join the broadcast; separated index groups move to the front)."""
    a.materialize()  # a pending (lazy) operand has no block yet
    adv_pos = [i for i, e in enumerate(entries) if e[0] in ("adv", "int")]
    idx_arrays = []
    for i in adv_pos:
        e = entries[i]
        idx_arrays.append(e[1] if e[0] == "adv" else None)
    bshape = ()
    for ix in idx_arrays:
        if ix is not None:
            try:
                bshape = _broadcast_shapes(bshape, ix.shape)
            except ValueError:
                raise IndexError("shape mismatch: indexing arrays could not be broadcast together with shapes "
                                 + " ".join(str(tuple(x.shape)) for x in idx_arrays if x is not None)) from None
    adjacent = builtins_all(entries[i][0] in ("adv", "int") for i in range(adv_pos[0], adv_pos[-1] + 1))
    # walk entries, collecting slice dims and per-index (axis extent, stride)
    off = a._offset
    pre, post = [], []  # (extent, src_stride) for slice/new dims before / after the index block
    idx_info = []
    ax = 0
    seen_adv = False
    for e in entries:
        kind = e[0]
        if kind == "new":
            (post if seen_adv else pre).append((1, 0))
            continue
        n, s = a.shape[ax], a._strides[ax]
        if kind == "slice":
            start, stop, step = e[1].indices(n)
            cnt = len(range(start, stop, step))
            off += start * s if cnt else 0
            (post if seen_adv else pre).append((cnt, s * step))
        elif kind == "int":
            i = e[1]
            if i < -n or i >= n:
                raise IndexError(f"index {i} is out of bounds for axis {ax} with size {n}")
            off += (i + n if i < 0 else i) * s
            seen_adv = True
        else:
            idx_info.append((e[1], n, s))
            seen_adv = True
        ax += 1
    if adjacent:
        dims = pre + [("b", n) for n in bshape] + post
        b0 = len(pre)
    else:
        dims = [("b", n) for n in bshape] + pre + post
        b0 = 0
    if len(dims) > MAX_NDIM:
        raise IndexError(f"indexing result would have {len(dims)} dimensions; the device supports {MAX_NDIM}")
    plan = IndexPlan()
    plan.ndim = len(dims)
    plan.n_idx = len(idx_info)
    out_shape = []
    for d, dim in enumerate(dims):
        if dim[0] == "b":
            plan.shape[d] = dim[1]
            plan.src_strides[d] = 0
            out_shape.append(dim[1])
        else:
            plan.shape[d] = dim[0]
            plan.src_strides[d] = dim[1]
            out_shape.append(dim[0])
    keep = []  # keep index arrays alive until the launch has been enqueued
    nb = len(bshape)
    for k, (ix, n, s) in enumerate(idx_info):
        if ix.dtype not in (np.dtype(np.int64), np.dtype(np.int32)):
            ix = astype(ix, np.int64)
        keep.append(ix)
        plan.idx_ptr[k] = ix.ptr
        plan.idx_dtype[k] = ix._code
        plan.idx_extent[k] = n
        plan.idx_mult[k] = s
        lead = nb - ix.ndim
        for j, (m, st) in enumerate(zip(ix.shape, ix._strides)):
            plan.idx_strides[k][b0 + lead + j] = 0 if (m == 1 and bshape[lead + j] != 1) else st
    base_ptr = a._block().ptr + off * a.dtype.itemsize
    return plan, tuple(out_shape), base_ptr, keep


def getitem(a, key):
    a = asarray(a)
    if isinstance(key, DeviceArray) and key.dtype.kind == "f":
        raise IndexError("arrays used as indices must be of integer (or boolean) type")
    entries, has_adv = _parse_key(a, key)
    if not has_adv:
        return _basic_view(a, entries)
    plan, out_shape, base_ptr, keep = _build_plan(a, entries)
    res = DeviceArray.empty(out_shape, a.dtype)
    if res.size:
        _lib().gather(plan, base_ptr, a._code, res.desc())
    del keep
    return res


def _scatter(a: DeviceArray, key, value, mode):
    _before_write(a)
    entries, has_adv = _parse_key(a, key)
    if not has_adv:
        dst = _basic_view(a, entries)
        if mode == _capi.SCATTER_SET:
            _copy_into(dst, value)
        else:
            if type(value) is int:
                value = _py_int_wrapped(dst.dtype, value)      # (ufunc.at casts its scalar operand)
            elif type(value) is float and dst.dtype.kind in "iu" and value == value and abs(value) < 2.0 ** 63:
                value = int(value)                             # (.. a float too: toward zero, then the integer add)
            _binary(np.add, _capi.B_ADD, dst, value, out=dst)
        return
    plan, out_shape, base_ptr, keep = _build_plan(a, entries)
    v = _operand(value)
    if isinstance(v, DeviceArray):
        if v.dtype != a.dtype:
            v = astype(v, a.dtype)
        # NumPy drops leading 1s of the value when broadcasting into the indexed shape
        while v.ndim > len(out_shape) and v.shape[0] == 1:
            v = v._view(v._offset, v.shape[1:], v._strides[1:])
        vd = v.desc(out_shape)
    else:
        if type(v) is int and mode == _capi.SCATTER_ADD:
            v = _py_int_wrapped(a.dtype, v)      # (ufunc.at casts its scalar operand; an assignment checks it)
        vd = _value_desc(v, a.dtype)
    if _prod(out_shape):
        _lib().scatter(plan, base_ptr, a._code, vd, mode)
    del keep


def setitem(a, key, value):
    _scatter(a, key, value, _capi.SCATTER_SET)


def index_add(a, indices, b=None):
    """np.add.at(a, indices, b): unbuffered, duplicates accumulate in index order
    (reference: minidiff/backend/numpy.py:105; used by definitions.py:186-189)."""
    if not isinstance(a, DeviceArray):
        raise TypeError("index_add needs a DeviceArray destination")
    _scatter(a, indices, b, _capi.SCATTER_ADD)


def _along_axis_plan(arr: DeviceArray, indices: DeviceArray, axis: int):
    if indices.dtype.kind not in "iu":
        raise IndexError("`indices` must be an integer array")
    if arr.ndim != indices.ndim:
        raise ValueError("`indices` and `arr` must have the same number of dimensions")
    if indices.dtype not in (np.dtype(np.int64), np.dtype(np.int32)):
        indices = astype(indices, np.int64)
    plan = IndexPlan()
    nd = arr.ndim
    plan.ndim = nd
    plan.n_idx = 1
    shape = []
    for d in range(nd):
        n = indices.shape[d]
        if d != axis:
            if arr.shape[d] != n and arr.shape[d] != 1 and n != 1:
                raise IndexError("shape mismatch: indexing arrays could not be broadcast together")
            n = builtins_max(n, arr.shape[d])
        shape.append(n)
    for d in range(nd):
        plan.shape[d] = shape[d]
        plan.src_strides[d] = 0 if (d == axis or arr.shape[d] == 1) else arr._strides[d]
        plan.idx_strides[0][d] = 0 if (indices.shape[d] == 1 and shape[d] != 1) else indices._strides[d]
    plan.idx_ptr[0] = indices.ptr
    plan.idx_dtype[0] = indices._code
    plan.idx_extent[0] = arr.shape[axis]
    plan.idx_mult[0] = arr._strides[axis]
    return plan, tuple(shape), indices


def take_along_axis(arr, indices, axis=-1):
    arr, indices = asarray(arr), asarray(indices)
    if axis is None:
        arr = ravel(arr)
        axis = 0
    ax = normalize_axis(axis, arr.ndim)
    plan, shape, keep = _along_axis_plan(arr, indices, ax)
    res = DeviceArray.empty(shape, arr.dtype)
    if res.size:
        _lib().gather(plan, arr.ptr, arr._code, res.desc())
    del keep
    return res


def put_along_axis(arr, indices, values, axis):
    if not isinstance(arr, DeviceArray):
        raise TypeError("put_along_axis needs a DeviceArray destination")
    indices = asarray(indices)
    _before_write(arr)
    if axis is None:
        if not arr.is_c_contiguous:
            raise ValueError("put_along_axis(axis=None) needs a contiguous destination")
        arr = reshape(arr, (-1,))
        axis = 0
    ax = normalize_axis(axis, arr.ndim)
    plan, shape, keep = _along_axis_plan(arr, indices, ax)
    v = _operand(values)
    if isinstance(v, DeviceArray):
        if v.dtype != arr.dtype:
            v = astype(v, arr.dtype)
        vd = v.desc(shape)
    else:
        vd = _value_desc(v, arr.dtype)
    if _prod(shape):
        _lib().scatter(plan, arr.ptr, arr._code, vd, _capi.SCATTER_SET)
    del keep


# ---- index utilities (device integer arithmetic) ---------------------------------------
def argwhere(a):
    a = asarray(a)
    if a.ndim == 0:      # (NumPy: one row of no coordinates for a non-zero scalar, none for zero)
        return zeros((1 if py_bool(a.item()) else 0, 0), dtype=np.int64)
    return stack(list(nonzero(a)), axis=1)


def isin(element, test_elements, assume_unique=False, invert=False, kind=None):
    """np.isin: `assume_unique` and `kind` only choose NumPy's algorithm (same answer); `invert` negates it."""
    if kind not in (None, "sort", "table"):
        raise ValueError(f"Invalid kind: '{kind}'. Please use None, 'sort' or 'table'.")
    e = asarray(element)
    t = ravel(asarray(test_elements))
    if t.size == 0:
        return full(e.shape, py_bool(invert), dtype=np.bool_)
    # element x test-element comparisons, the test elements in slabs that keep the boolean intermediate under 1 GiB
    slab = builtins_max(1, (1 << 30) // builtins_max(e.size, 1))
    res = None
    for c0 in range(0, t.size, slab):
        part = t[c0:c0 + slab]
        hit = any(equal(expand_dims(e, e.ndim), reshape(part, (1,) * e.ndim + (part.size,))), axis=e.ndim)
        res = hit if res is None else logical_or(res, hit)
    return logical_not(res) if invert else res


def unravel_index(indices, shape, order="C"):
    if order == "F":      # column-major: the C decomposition over the reversed shape, coordinates handed back in axis order
        return tuple(reversed(unravel_index(indices, tuple(reversed(_normalize_shape(shape))), "C")))
    if order != "C":
        raise ValueError("only 'C' or 'F' order is permitted")
    idx = asarray(indices)
    if idx.dtype.kind not in "iu":
        raise TypeError("only int indices permitted")
    shape = _normalize_shape(shape)
    total = _prod(shape)
    if idx.size and (py_bool(any(less(idx, 0)).item()) or py_bool(any(greater_equal(idx, total)).item())):
        raise ValueError(f"index is out of bounds for array with size {total}")
    return _unravel(idx if idx.dtype == np.int64 else astype(idx, np.int64), shape)


def materialize(a):
    """Force a pending (lazy) array into HBM; a no-op for materialised arrays."""
    if isinstance(a, DeviceArray) and a._buf is None:
        a._materialize()
    return a


def materialize_many(arrays):
    """Evaluate several pending arrays, sharing one pass among those of equal shape and float
    type (gradients of one backward sweep read the same operands: each distinct leaf is then
    loaded once and all results are stored by the same kernel, `mdhip_vm_eval_multi`)."""
    pending, seen = [], set()
    for a in arrays:
        if isinstance(a, DeviceArray) and a._expr is not None and id(a) not in seen:
            seen.add(id(a))
            pending.append(a)
    groups = {}
    for a in pending:
        if a._expr.kind != _lz.GEMM and a.size and a.dtype in _FLOAT_DT and _FLOAT_DT[a.dtype] == a._expr.cdt and not a._tasks:  # (owed reductions ride on a single evaluation)
            groups.setdefault((a.shape, a._expr.cdt), []).append(a)
    for group in groups.values():
        while len(group) >= 2:
            batch, leaves = [], {}
            for a in group:
                merged = dict(leaves)
                merged.update(a._expr.leaves)
                if len(batch) < 4 and len(merged) <= _lz.MAX_LEAVES:
                    batch.append(a)
                    leaves = merged
            taken = {id(b) for b in batch}
            group = [a for a in group if id(a) not in taken]
            if len(batch) < 2:
                continue
            progs = (_capi.VmProgram * len(batch))()
            outs = (_capi.ArrayDesc * len(batch))()
            keep = []
            saved = [a._expr for a in batch]
            for k, a in enumerate(batch):
                prog, kp = _lz.build_program(a._expr, a.shape)
                progs[k] = prog
                keep.append(kp)
                a._buf = _Buffer(_prod(a.shape) * a.dtype.itemsize)
                a._expr = None
                outs[k] = a.desc()
            try:
                _lib().vm_eval_multi(progs, outs, len(batch))
            except BaseException:
                for a, e in zip(batch, saved):  # nothing was written: stay pending
                    a._buf, a._expr, a._cdesc = None, e, None
                raise
            FUSION_STATS["vm_eval_multi"] += 1
            del keep
    for a in pending:
        a.materialize()
    for a in arrays:  # results whose fill is still owed (deferred reductions)
        if isinstance(a, DeviceArray) and a._buf is not None and a._buf.task is not None:
            a._buf.task.run()
    return arrays


def synchronize():
    _lib().sync()


# ---- opt-in device RNG (csrc/md_rng.h; reference aliases: backend/numpy.py:131-137) ------------------------------------------
# Default OFF: the reference's random functions ARE np.random.*, so a given np.random.seed must give NumPy's numbers, which
# only the host can produce (hip_backend draws there and uploads). MDHIP_DEVICE_RNG=1 / device_rng(True, seed) switches
# rand / randn / randint / binomial / permutation / shuffle / choice to a counter-based generator on the device: same
# distributions, its own stream (reproducible per seed, identical on the CPU test double), nothing crosses PCIe.
class _DeviceRng:
    enabled = os.environ.get("MDHIP_DEVICE_RNG") == "1"
    seed = int(os.environ.get("MDHIP_DEVICE_RNG_SEED", "0")) & 0xFFFFFFFFFFFFFFFF
    offset = 0          # Philox blocks consumed so far


def device_rng(enable: bool = True, seed=None) -> bool:
    """Switch the device generator on / off (returns the previous state); `seed` restarts its stream."""
    prev = _DeviceRng.enabled
    _DeviceRng.enabled = bool(enable)
    if seed is not None:
        _DeviceRng.seed, _DeviceRng.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, 0
    return prev


def device_rng_enabled() -> bool:
    return _DeviceRng.enabled


def _rng_take(words: int) -> int:
    off = _DeviceRng.offset
    _DeviceRng.offset += (int(words) + 3) // 4
    return off


_RNG_UNIFORM, _RNG_NORMAL, _RNG_INTEGERS, _RNG_BINOMIAL = 0, 1, 2, 3


def _random_fill(kind, shape, dtype, a=0.0, b=0.0, words_per_elem=1):
    res = DeviceArray.empty(shape, dtype)
    if res.size:
        _lib().random_fill(kind, _DeviceRng.seed, _rng_take(res.size * words_per_elem), float(a), float(b), res.desc())
    return res


def random_uniform(shape, dtype=np.float64):
    dtype = np.dtype(dtype)
    return _random_fill(_RNG_UNIFORM, shape, dtype, words_per_elem=2 if dtype == np.float64 else 1)


def random_normal(shape, dtype=np.float64):
    dtype = np.dtype(dtype)
    return _random_fill(_RNG_NORMAL, shape, dtype, words_per_elem=4 if dtype == np.float64 else 2)


def random_integers(low, high, shape, dtype=np.int64):
    low, high = operator.index(low), operator.index(high)
    if high <= low:
        raise ValueError("low >= high")
    if high - low > (1 << 53):
        raise ValueError("device RNG: integer ranges up to 2**53")
    info = np.iinfo(np.dtype(dtype))   # NumPy's own bounds check (np.random.randint): no silent wrap into a narrower dtype
    if low < info.min:
        raise ValueError(f"low is out of bounds for {np.dtype(dtype).name}")
    if high - 1 > info.max:
        raise ValueError(f"high is out of bounds for {np.dtype(dtype).name}")
    return _random_fill(_RNG_INTEGERS, shape, np.dtype(dtype), low, high - low, words_per_elem=2)


RANDOM_BINOMIAL_MAX_N = 256


def random_binomial(n, p, shape, dtype=np.int64):
    n = operator.index(n)
    if n < 0 or n > RANDOM_BINOMIAL_MAX_N:
        raise ValueError(f"device RNG: binomial with 0 <= n <= {RANDOM_BINOMIAL_MAX_N}")
    if not (0.0 <= float(p) <= 1.0):          # (also rejects nan)
        raise ValueError("p < 0, p > 1 or p is NaN")
    return _random_fill(_RNG_BINOMIAL, shape, np.dtype(dtype), n, float(p), words_per_elem=_bi.max(n, 1))


def random_permutation(n: int):
    """A uniformly random permutation of 0..n-1 (int64): the indices sorted by a 64-bit random key each, on the device."""
    n = operator.index(n)
    if n < 0:
        raise ValueError("negative dimensions are not allowed")
    res = DeviceArray.empty((n,), np.int64)
    if n:
        _lib().random_permutation(_DeviceRng.seed, _rng_take(2 * n), res.desc())
    return res


# storage-only dtypes (float16, int8/16, uint8/16/32/64): the computing functions above are wrapped so that a call with such an
# operand runs promote -> wide kernel -> demote; calls on the compute dtypes pay one flag test per argument (narrow.py)
_narrow.install(globals())


# ---- the C route in front of the float elementwise entries (csrc/fastpath.c) -------------------------------------------------
def _install_fastpath(ns):
    if _fp is None:
        return

    def raise_status(st):
        lib = _lib()
        raise _capi._EXC.get(st, RuntimeError)(lib.cdll.mdhip_last_error().decode(errors="replace"))

    by_code = {c: dt for dt, c in _DTYPE_CODES.items() if c < _NARROW_MIN}
    _fp.configure(DeviceArray, tuple(by_code[c] for c in range(_NARROW_MIN)), dtype_code, raise_status, _lib)
    _fp.set_lazy(_LAZY)

    def bind(lib):
        _fp.bind({name: _capi.entry_address(lib, name) for name in ("mdhip_alloc", "mdhip_free", "mdhip_unary", "mdhip_binary", "mdhip_reduce", "mdhip_matmul",
                                                                     "mdhip_where")})
        _fp.enable_ops(True)

    _capi._BIND_HOOKS.append(bind)
    if _capi._LIB is not None:
        bind(_capi._LIB)
    unary = {"absolute": _capi.U_ABS, "negative": _capi.U_NEG, "sign": _capi.U_SIGN, "ceil": _capi.U_CEIL, "floor": _capi.U_FLOOR,
             "sin": _capi.U_SIN, "cos": _capi.U_COS, "tan": _capi.U_TAN, "sinh": _capi.U_SINH, "cosh": _capi.U_COSH, "tanh": _capi.U_TANH,
             "exp": _capi.U_EXP, "log": _capi.U_LOG, "sqrt": _capi.U_SQRT, "logical_not": _capi.U_LOGICAL_NOT, "isnan": _capi.U_ISNAN}
    binary = {"add": _capi.B_ADD, "subtract": _capi.B_SUB, "multiply": _capi.B_MUL, "true_divide": _capi.B_TRUE_DIV,
              "floor_divide": _capi.B_FLOOR_DIV, "mod": _capi.B_MOD, "power": _capi.B_POW, "maximum": _capi.B_MAXIMUM,
              "minimum": _capi.B_MINIMUM, "equal": _capi.B_EQ, "not_equal": _capi.B_NE, "less": _capi.B_LT, "less_equal": _capi.B_LE,
              "greater": _capi.B_GT, "greater_equal": _capi.B_GE}
    for name, code in unary.items():
        ns[name] = _fp.FastOp(1, code, ns[name], name)
    for name, code in binary.items():
        ns[name] = _fp.FastOp(2, code, ns[name], name)
    ns["matmul"] = _fp.FastOp(3, 0, ns["matmul"], "matmul")
    ns["where"] = _fp.FastOp(4, 0, ns["where"], "where")
    for name, code in (("sum", _capi.R_SUM), ("prod", _capi.R_PROD), ("max", _capi.R_MAX), ("min", _capi.R_MIN)):
        ns[name] = _fp.FastOp(5, code, ns[name], name)


_install_fastpath(globals())
