"""Shim (DeviceArray + NumPy-semantics layer over the C-ABI) vs NumPy.

The same cases run twice: on the CPU test double here (`-m "not gpu"`, proves
the host logic) and on the HIP library on a real MI355X (`-m gpu`, the parity
tests proper — every call goes through libmdhip.so)."""
import numpy as np
import pytest

import cases_shim as cs

FLOATS = [np.float32, np.float64]
NUMS = [np.float32, np.float64, np.int64, np.int32]
MIXED = ["add", "multiply", "true_divide", "less", "power", "maximum", "logical_or"]
RDT = [np.float32, np.float64, np.int64]
MDT = [np.float32, np.float64, np.int64]
gpu = pytest.mark.gpu


def _run(lib, on_gpu, want_gpu, fn, *args):
    from minidiff_amd import ndarray
    if want_gpu:
        assert on_gpu and lib.target == "hip:gfx950"
    elif on_gpu:
        pytest.skip("GPU present: covered by the gpu-marked twin")
    fn(ndarray, *args)


@pytest.mark.parametrize("dtype", NUMS)
@pytest.mark.parametrize("name", cs.BINARY)
def test_binary_cpu(lib, on_gpu, name, dtype): _run(lib, on_gpu, False, cs.case_binary, name, dtype)
@gpu
@pytest.mark.parametrize("dtype", NUMS)
@pytest.mark.parametrize("name", cs.BINARY)
def test_binary_gpu(lib, on_gpu, name, dtype): _run(lib, on_gpu, True, cs.case_binary, name, dtype)

@pytest.mark.parametrize("name", MIXED)
def test_binary_mixed_cpu(lib, on_gpu, name): _run(lib, on_gpu, False, cs.case_binary_mixed, name)
@gpu
@pytest.mark.parametrize("name", MIXED)
def test_binary_mixed_gpu(lib, on_gpu, name): _run(lib, on_gpu, True, cs.case_binary_mixed, name)

@pytest.mark.parametrize("dtype", FLOATS)
@pytest.mark.parametrize("name", cs.UNARY_F)
def test_unary_cpu(lib, on_gpu, name, dtype): _run(lib, on_gpu, False, cs.case_unary, name, dtype)
@gpu
@pytest.mark.parametrize("dtype", FLOATS)
@pytest.mark.parametrize("name", cs.UNARY_F)
def test_unary_gpu(lib, on_gpu, name, dtype): _run(lib, on_gpu, True, cs.case_unary, name, dtype)

@pytest.mark.parametrize("dtype", RDT)
@pytest.mark.parametrize("name", cs.REDUCE)
def test_reduce_cpu(lib, on_gpu, name, dtype): _run(lib, on_gpu, False, cs.case_reduce, name, dtype)
@gpu
@pytest.mark.parametrize("dtype", RDT)
@pytest.mark.parametrize("name", cs.REDUCE)
def test_reduce_gpu(lib, on_gpu, name, dtype): _run(lib, on_gpu, True, cs.case_reduce, name, dtype)

@pytest.mark.parametrize("dtype", MDT)
def test_matmul_cpu(lib, on_gpu, dtype): _run(lib, on_gpu, False, cs.case_matmul, dtype)
@gpu
@pytest.mark.parametrize("dtype", MDT)
def test_matmul_gpu(lib, on_gpu, dtype): _run(lib, on_gpu, True, cs.case_matmul, dtype)

SINGLE = ["case_unary_int", "case_reduce_large", "case_argreduce", "case_layout", "case_matmul_mfma", "case_where_clip",
          "case_indexing", "case_index_utils", "case_inplace", "case_errors"]

@pytest.mark.parametrize("case", SINGLE)
def test_case_cpu(lib, on_gpu, case): _run(lib, on_gpu, False, getattr(cs, case))
@gpu
@pytest.mark.parametrize("case", SINGLE)
def test_case_gpu(lib, on_gpu, case): _run(lib, on_gpu, True, getattr(cs, case))


def _overlap_case(nd_unused):
    """`a[1:] += a[:-1]`, `a[1:] = a[:-1]`, `a += a.T`, `a[::-1] += a`: NumPy gives the result of
    operating on a copy of the overlapping operand (tensor.py:269-362 hands raw arrays to these)."""
    import numpy as np
    from minidiff_amd import ndarray as nd
    for dt in (np.float64, np.float32, np.int64):
        h = np.arange(40, dtype=dt)
        d = nd.asarray(h)
        h[1:] += h[:-1]
        d[1:] += d[:-1]
        assert np.array_equal(np.asarray(d), h)
        h[1:] = h[:-1]
        d[1:] = d[:-1]
        assert np.array_equal(np.asarray(d), h)
        h[::-1] += h
        d[::-1] += d
        assert np.array_equal(np.asarray(d), h)
        h2 = np.arange(36, dtype=dt).reshape(6, 6)
        d2 = nd.asarray(h2)
        h2 += h2.T
        d2 += d2.T
        assert np.array_equal(np.asarray(d2), h2)
        h2 *= h2
        d2 *= d2                       # the identical view: no copy needed, same answer
        assert np.array_equal(np.asarray(d2), h2)
        h2[:, 1:] -= h2[:, :-1]
        d2[:, 1:] -= d2[:, :-1]
        assert np.array_equal(np.asarray(d2), h2)
        h2[::2] = h2[1::2]             # interleaved rows: disjoint, but ranges intersect (copied, still right)
        d2[::2] = d2[1::2]
        assert np.array_equal(np.asarray(d2), h2)


def test_inplace_overlap_cpu(lib, on_gpu): _run(lib, on_gpu, False, _overlap_case)
@gpu
def test_inplace_overlap_gpu(lib, on_gpu): _run(lib, on_gpu, True, _overlap_case)


def test_integer_power_with_negative_exponents(lib):
    """NumPy raises "Integers to negative integer powers are not allowed" from INSIDE its loop: a non-empty result raises (array or
    scalar exponent), an EMPTY one never meets the exponent and is returned (fuzz seed 92, case 5327)."""
    from minidiff_amd import ndarray as nd
    base = nd.asarray(np.arange(1, 7, dtype=np.int64).reshape(2, 3))
    neg = nd.asarray(np.array([1, -2, 3], dtype=np.int32))
    with pytest.raises(ValueError, match="negative integer powers"):
        nd.power(base, neg)
    with pytest.raises(ValueError, match="negative integer powers"):
        nd.power(base, -3)
    empty = nd.asarray(np.zeros((0, 2, 3), dtype=np.int64))
    for e in (neg, -3):
        got = nd.power(empty, e)
        exp = np.power(empty.get(), e.get() if isinstance(e, nd.DeviceArray) else e)
        assert got.shape == exp.shape and got.dtype == exp.dtype
    np.testing.assert_array_equal(nd.power(base, nd.asarray(np.array([0, 2, 3], dtype=np.int32))).get(), np.power(base.get(), np.array([0, 2, 3], dtype=np.int32)))


def test_keywords_are_honoured_or_refused_never_dropped(lib):
    """NumPy keywords of the table's functions: `out=` lands the result in the caller's array (ufuncs, reductions, mean / std) under
    NumPy's shape and same-kind rules; a keyword this backend has no code for is accepted at its default and raises TypeError at any
    other value — never silently ignored."""
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((4, 5)).astype(np.float32), rng.standard_normal((4, 5)).astype(np.float32)
    da, db = nd.asarray(a), nd.asarray(b)
    for name in ("add", "multiply", "less", "maximum"):
        o64, r64 = nd.zeros((4, 5), np.float64), np.zeros((4, 5), np.float64)
        ret = getattr(nd, name)(da, db, out=o64)
        getattr(np, name)(a, b, out=r64)
        assert ret is o64 and np.array_equal(o64.get(), r64), name
        ret = getattr(nd, name)(da, db, out=(o64,), where=True, casting="same_kind", dtype=None)
        assert ret is o64
    o, r = nd.zeros((4, 5), np.float32), np.zeros((4, 5), np.float32)
    assert nd.sin(da, out=o) is o
    np.sin(a, out=r)
    assert np.allclose(o.get(), r, rtol=1e-6)
    with pytest.raises(np._core._exceptions._UFuncOutputCastingError):
        nd.add(da, db, out=nd.zeros((4, 5), np.int32))
    with pytest.raises(ValueError, match="non-broadcastable"):
        nd.add(da, db, out=nd.zeros((5, 4), np.float32))
    with pytest.raises(ValueError, match="non-broadcastable"):
        nd.sin(da, out=nd.zeros((5, 4), np.float32))
    for name in ("sum", "prod", "max", "min", "mean", "std", "argmax", "any"):
        ref = getattr(np, name)(a, axis=0)
        o = nd.zeros(ref.shape, ref.dtype)
        ret = getattr(nd, name)(da, axis=0, out=o)
        assert ret is o and np.allclose(o.get(), ref, rtol=1e-5), name
        assert getattr(da, name)(axis=0, out=o) is o
        with pytest.raises(ValueError, match="wrong shape"):
            getattr(nd, name)(da, axis=0, out=nd.zeros((4,), ref.dtype))
    with pytest.raises(TypeError, match="Cannot cast"):
        nd.sum(da, axis=0, out=nd.zeros((5,), np.int32))
    # unsupported keywords: loud
    for call in (lambda: nd.add(da, db, where=nd.asarray(a > 0)), lambda: nd.add(da, db, dtype=np.float64), lambda: nd.sum(da, where=nd.asarray(a > 0)),
                 lambda: nd.sum(da, initial=3.0), lambda: nd.max(da, initial=0.0), lambda: nd.zeros((2, 2), order="F"), lambda: nd.add(da, db, casting="unsafe"),
                 lambda: nd.concatenate([da, db], dtype=np.float64), lambda: nd.std(da, correction=1), lambda: nd.sin(da, where=False)):
        with pytest.raises(TypeError, match="not supported"):
            call()
    with pytest.raises(TypeError, match="unexpected keyword"):
        nd.add(da, db, axis=0)
    # .. at their defaults they pass
    assert np.array_equal(nd.sum(da, axis=1, where=True, initial=None).get(), a.sum(axis=1))
    assert nd.zeros((2, 2), order="C", like=None).shape == (2, 2) and da.astype(np.float64, casting="unsafe", order="K").dtype == np.float64


def test_memory_order_arguments_follow_numpy(lib):
    """`order=` of ravel / flatten / copy / reshape / unravel_index on views of every kind (transposed, permuted, sliced, reversed,
    broadcast): the VALUES NumPy returns — 'A' and 'K' depend on the view's strides — and, for copies, NumPy's result layout (a 'K'
    copy of x.T is laid out like x.T). isin's and the *_like functions' keywords."""
    from minidiff_amd import ndarray as nd
    a = np.arange(24.).reshape(2, 3, 4)
    d = nd.asarray(a)
    views = {"T": (d.T, a.T), "swap": (nd.swapaxes(d, 0, 1), np.swapaxes(a, 0, 1)), "slice": (d[:, ::2, 1:], a[:, ::2, 1:]),
             "perm": (nd.transpose(d, (1, 2, 0)), np.transpose(a, (1, 2, 0))), "neg": (d[::-1].T, a[::-1].T), "plain": (d, a),
             "bcast": (nd.broadcast_to(d[:, :1, :], (2, 3, 4)), np.broadcast_to(a[:, :1, :], (2, 3, 4)))}

    def same(got, exp, what, layout=False):
        assert got.dtype == exp.dtype and np.array_equal(got.get(), exp), what
        if layout:
            assert got.is_c_contiguous == exp.flags.c_contiguous and got.T.is_c_contiguous == exp.flags.f_contiguous, what

    for vn, (dv, av) in views.items():
        for order in "CFAK":
            same(nd.ravel(dv, order=order), np.ravel(av, order=order), ("ravel", vn, order))
            same(nd.flatten(dv, order=order), av.flatten(order=order), ("flatten", vn, order))
            same(nd.copy(dv, order=order), np.copy(av, order=order), ("copy", vn, order), layout=True)
            same(dv.copy(order=order), av.copy(order=order), ("method copy", vn, order), layout=True)
            if order != "K":
                same(nd.reshape(dv, (av.size // 6, 6), order=order), np.reshape(av, (av.size // 6, 6), order=order), ("reshape", vn, order))
        same(nd.copy(dv), np.copy(av), ("copy default", vn), layout=True)        # np.copy: 'K'
        same(dv.copy(), av.copy(), ("method copy default", vn), layout=True)     # ndarray.copy: 'C'
    with pytest.raises(ValueError):
        nd.reshape(d, (4, 6), order="K")
    with pytest.raises(ValueError):
        nd.ravel(d, order="Z")
    idx = np.array([5, 7, 23, 0])
    for order in "CF":
        for g, e in zip(nd.unravel_index(nd.asarray(idx), (2, 3, 4), order=order), np.unravel_index(idx, (2, 3, 4), order=order)):
            assert np.array_equal(g.get(), e), order
    with pytest.raises(ValueError):
        nd.unravel_index(nd.asarray(idx), (2, 3, 4), order="K")
    te = np.array([1., 5., 30.])
    for kw in ({}, {"invert": True}, {"assume_unique": True}, {"kind": "sort"}, {"invert": True, "assume_unique": True}):
        same(nd.isin(d, nd.asarray(te), **kw), np.isin(a, te, **kw), ("isin", kw))
    same(nd.isin(d, nd.asarray(np.array([])), invert=True), np.isin(a, np.array([]), invert=True), "isin empty")
    with pytest.raises(ValueError):
        nd.isin(d, nd.asarray(te), kind="bogus")
    for f in ("zeros_like", "ones_like"):
        same(getattr(nd, f)(d, shape=(2, 2)), getattr(np, f)(a, shape=(2, 2)), f)
        same(getattr(nd, f)(d, subok=False, dtype=np.int32), getattr(np, f)(a, subok=False, dtype=np.int32), f)
    same(nd.full_like(d, 3, shape=5), np.full_like(a, 3, shape=5), "full_like")
