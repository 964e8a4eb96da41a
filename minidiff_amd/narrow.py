"""Storage-only dtypes: float16, int8/16, uint8/16/32/64 — the names of the reference table
(minidiff/backend/numpy.py:188-200) beyond the five the kernels compute in.

Arrays of these types live in device memory like any other (views, transfers, `astype`, strided copies: one conversion
kernel, `mdhip_convert`). Elementwise arithmetic, `where` / `clip` and the reductions on them run NATIVELY since round 4 —
one launch that loads each operand in its own type, computes in the loop dtype's carrier (int32 / int64 / uint64 /
float32: csrc/md_narrow.h) and stores in the result's type, 1.0x the algorithmic traffic — and so do the gathers / scatters
by index arrays (csrc/index.hip: elements move by size; np.add.at wraps the integers and rounds float16 after every
contribution, in index order, as NumPy's unbuffered loop does). The functions listed in COMPUTE
below still go

    promote to a wide device type  ->  the ordinary kernel  ->  demote to NumPy's result dtype

which is what NumPy's own float16 loops do (compute in float32, round once), and exact for the integers: the low bits
of a sum / difference / product / power computed in 64 bits are the narrow result, so NumPy's wrap-around falls out of
the truncating conversion; comparisons, divisions and reductions see the true values. NumPy's result dtype (and the
exception it raises for a combination it rejects) comes from running the SAME NumPy function on one-element host
dummies of the operands' dtypes — no promotion table restated here.

In the promote path uint64 rides in int64 with the same bits: a wrapped function that looks at VALUES (not just bits)
first checks that no element is >= 2**63 and raises TypeError otherwise — loud, not wrong. No function of the table is left
that does: the products and the nonzero family give the same bits either way (_BITS), a call whose NumPy loop is float64
(uint64 @ int64, mean / std of integers) converts each element straight from its own type, `isin` is composed of the native
comparisons, and the native path has unsigned 64-bit loops.

Nothing on a BASELINE path uses these types; the cost on the wide paths is one flag test per call (`install`)."""
from __future__ import annotations

import numpy as np

_WIDE = {
    np.dtype(np.int8): np.dtype(np.int64), np.dtype(np.int16): np.dtype(np.int64),
    np.dtype(np.uint8): np.dtype(np.int64), np.dtype(np.uint16): np.dtype(np.int64), np.dtype(np.uint32): np.dtype(np.int64),
    np.dtype(np.uint64): np.dtype(np.int64),   # same bits (range-checked where values matter)
    np.dtype(np.float16): np.dtype(np.float32),
}
_U64 = np.dtype(np.uint64)
NARROW_CODE_MIN = 5   # _capi.I8: DeviceArray._code >= this <=> storage-only dtype

# functions that move elements without looking at their values: uint64 needs no range check there
_MOVERS = {"concatenate", "stack", "tile", "repeat", "split"}
# the statistics: NumPy's own first step is float64 (the sum runs in float64, each element converted first), so uint64 widens
# straight to float64 there — the whole range, no detour through int64
_STATS = {"mean", "std"}
# functions whose integer results are the same BITS whether 64-bit operands are read signed or unsigned (sums of products wrap
# mod 2**64 either way; "is it zero" does not look at the sign): uint64 rides in int64 with no range check there
_BITS = {"matmul", "dot", "tensordot", "nonzero", "flatnonzero", "argwhere"}

# Functions whose C entry points take the storage-only dtypes DIRECTLY (one launch, each operand read in its own type, the result
# written in its own: csrc/narrow.hip, the 12-dtype loads of the reduction kernels) — every elementwise ufunc, where / clip and the
# reductions — are NOT wrapped. What is wrapped below (promote -> wide kernel -> demote) are the functions that have no kernel for
# these types yet: products, the composed statistics, the array builders. (Gathers and scatters by index arrays — getitem, setitem,
# index_add, take_/put_along_axis — move or add in the array's own type: csrc/index.hip.)
COMPUTE = [
    "mean", "std", "matmul", "dot", "tensordot",
    "concatenate", "stack", "tile", "repeat", "split", "nonzero", "flatnonzero", "argwhere",
]


def is_narrow_dtype(dt) -> bool:
    return dt is not None and np.dtype(dt) in _WIDE


def install(ns: dict):
    """Wrap the computing functions of minidiff_amd.ndarray (its module globals `ns`)."""
    DeviceArray = ns["DeviceArray"]

    def has_narrow(x) -> bool:
        if type(x) is DeviceArray:
            return x._code >= NARROW_CODE_MIN
        if isinstance(x, (np.generic, np.ndarray)):
            return x.dtype in _WIDE
        if isinstance(x, (list, tuple)):
            for y in x:
                if has_narrow(y):
                    return True
        return False

    def any_narrow(args, kw) -> bool:
        for x in args:
            if has_narrow(x):
                return True
        if kw:
            for k, v in kw.items():
                if k == "dtype":
                    if v is not None and is_narrow_dtype(v):
                        return True
                elif has_narrow(v):
                    return True
        return False

    convert = ns["_convert"]

    def dummy(x):
        if type(x) is DeviceArray or isinstance(x, np.ndarray):
            return np.ones(tuple(1 if s else 0 for s in x.shape), dtype=x.dtype)
        if isinstance(x, (list, tuple)):
            return type(x)(dummy(y) for y in x)
        return x

    def widen(x, name, to_f64=False):
        """`to_f64`: NumPy's loop for this call is float64 although the narrow operand is an integer (uint64 @ int64, mean of
        uint64): each element is converted first — straight from its own type, the whole uint64 range."""
        if type(x) is DeviceArray:
            if x._code >= NARROW_CODE_MIN:
                if to_f64 and x.dtype.kind in "iu":
                    return convert(x, np.dtype(np.float64))
                w = convert(x, _WIDE[x.dtype])
                if x.dtype == _U64 and name not in _MOVERS and name not in _BITS and name not in _STATS and x.size and bool(ns["any"](ns["less"](w, 0)).item()):
                    raise TypeError(f"uint64 values >= 2**63 are not supported by the MI355X backend in {name}()")
                return w
            return x
        if isinstance(x, np.generic) and x.dtype in _WIDE:
            if to_f64 and x.dtype.kind in "iu":
                return np.float64(x)
            if x.dtype == _U64 and name not in _BITS and int(x) >= 1 << 63:
                raise TypeError(f"uint64 values >= 2**63 are not supported by the MI355X backend in {name}()")
            return np.asarray(x).astype(_WIDE[x.dtype])[()]      # (uint64 >= 2**63 under _BITS: the same bits)
        if isinstance(x, np.ndarray) and x.dtype in _WIDE:
            if to_f64 and x.dtype.kind in "iu":
                return x.astype(np.float64)
            if x.dtype == _U64 and name not in _MOVERS and name not in _BITS and name not in _STATS and x.size and int(x.max()) >= 1 << 63:
                raise TypeError(f"uint64 values >= 2**63 are not supported by the MI355X backend in {name}()")
            return x.astype(_WIDE[x.dtype])
        if isinstance(x, (list, tuple)):
            return type(x)(widen(y, name, to_f64) for y in x)
        return x

    def demote(res, ref):
        """`res`: what the wide call returned; `ref`: what NumPy returned for the dummies (same structure)."""
        if type(res) is DeviceArray:
            rdt = ref.dtype if isinstance(ref, (np.ndarray, np.generic)) else None
            if rdt is not None and rdt != res.dtype:
                return convert(res, rdt)
            return res
        if isinstance(res, (list, tuple)) and isinstance(ref, (list, tuple)) and len(res) == len(ref):
            return type(res)(demote(r, f) for r, f in zip(res, ref))
        return res

    def make(name, fn):
        npf = getattr(np, name)

        def wrapped(*args, **kw):
            for x in args:                         # the common case inline: device arrays of the compute dtypes and Python scalars
                tx = type(x)
                if tx is DeviceArray:
                    if x._code >= NARROW_CODE_MIN:
                        break
                elif tx is int or tx is float or tx is bool or x is None:
                    continue
                elif has_narrow(x):
                    break
            else:
                if not kw or not any_narrow((), kw):
                    return fn(*args, **kw)
            # NumPy's own verdict on dtypes (and its exceptions) from one-element dummies
            dargs = [dummy(a) for a in args]
            dkw = {k: dummy(v) for k, v in kw.items()}
            with np.errstate(all="ignore"):
                ref = npf(*dargs, **dkw)
            f64 = (name in _STATS or name in _BITS) and isinstance(ref, (np.ndarray, np.generic)) and ref.dtype == np.float64
            wargs = [widen(a, name, f64) for a in args]
            wkw = {k: (_WIDE[np.dtype(v)] if k == "dtype" and v is not None and is_narrow_dtype(v) else widen(v, name, f64)) for k, v in kw.items()}
            return demote(fn(*wargs, **wkw), ref)

        wrapped.__name__ = getattr(fn, "__name__", name)
        wrapped.__doc__ = getattr(fn, "__doc__", None)
        wrapped.__wrapped__ = fn
        return wrapped

    for name in COMPUTE:
        if name in ns:
            ns[name] = make(name, ns[name])

    def inplace(ufunc, fn, a, b):
        """a OP= b for a narrow `a` (or a wide `a` with a narrow `b`): NumPy's casting verdict from the dummies, the
        arithmetic in the wide type, the result converted back into a's memory."""
        da, db = dummy(a), dummy(b)
        with np.errstate(all="ignore"):
            ufunc(da, db, out=da)                   # raises NumPy's UFuncTypeError where NumPy would
        res = fn(widen(a, ufunc.__name__), widen(b, ufunc.__name__))
        ns["_copy_into"](a, res)
        return a

    ns["_narrow_inplace"] = inplace
    ns["_any_narrow"] = any_narrow
