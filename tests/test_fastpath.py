"""The C route of eager backend calls (minidiff_amd/csrc/fastpath.c): what it serves, that it gives the
Python implementation's answers bit for bit, what it hands back to Python, and that its objects behave like the ones
they replace. Runs on whichever library the session is bound to (the CPU double here, libmdhip on the GPU box)."""
import gc
import weakref
import ctypes as C

import numpy as np
import pytest

from minidiff_amd import ndarray as nd

fp = nd._fp
pytestmark = pytest.mark.skipif(fp is None, reason="MDHIP_FASTPATH=0 / MDHIP_TRACE: pure-Python host path")

BINARY = ["add", "subtract", "multiply", "true_divide", "floor_divide", "mod", "power", "maximum", "minimum",
          "equal", "not_equal", "less", "less_equal", "greater", "greater_equal"]
UNARY = ["absolute", "negative", "sign", "ceil", "floor", "sin", "cos", "tan", "sinh", "cosh", "tanh", "exp", "log", "sqrt",
         "logical_not", "isnan"]


def _served(fn, *args):
    before = fp.stats()
    out = fn(*args)
    after = fp.stats()
    return out, after["served"] - before["served"], after["passed"] - before["passed"]


def _same(a, b):
    assert type(a) is type(b) is nd.DeviceArray
    assert a.dtype == b.dtype and a.shape == b.shape and a._strides == b._strides and a._code == b._code
    np.testing.assert_array_equal(a.get(), b.get())


@pytest.fixture()
def eager(lib):
    prev = nd.set_lazy(False)
    yield
    nd.set_lazy(prev)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_binary_c_route_equals_python_route(eager, dtype):
    rng = np.random.default_rng(3)
    base = rng.standard_normal((6, 5, 4)).astype(dtype) + 2.5      # positive enough for power / log-free ops to be finite mostly
    A = nd.asarray(base)
    operands = [
        (A, A), (A, nd.asarray(base[0])), (nd.asarray(base[:, :1, :]), A), (A, nd.asarray(base[0, 0])),
        (A[::2], A[1::2]), (A[:, ::-1], A), (A[..., 1:3], A[..., 0:2]), (nd.asarray(base[0, 0, 0].reshape(())), A),
        (A, 2), (3, A), (A, 0.5), (-1.25, A), (A[:0], A[:0]), (nd.asarray(base.reshape(-1)[:7]), 2.0),
    ]
    with np.errstate(all="ignore"):
        for name in BINARY:
            fast = getattr(nd, name)
            assert type(fast) is fp.FastOp and fast.__name__ == name
            for a, b in operands:
                got, served, passed = _served(fast, a, b)
                assert (served, passed) == (1, 0), (name, getattr(a, "shape", a), getattr(b, "shape", b))
                _same(got, fast.__wrapped__(a, b))
                exp = getattr(np, name if name != "mod" else "remainder")(a.get() if isinstance(a, nd.DeviceArray) else a,
                                                                         b.get() if isinstance(b, nd.DeviceArray) else b)
                assert got.dtype == exp.dtype and got.shape == exp.shape


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_unary_c_route_equals_python_route(eager, dtype):
    rng = np.random.default_rng(4)
    base = (rng.standard_normal((5, 7)) * 2).astype(dtype)
    A = nd.asarray(base)
    with np.errstate(all="ignore"):
        for name in UNARY:
            fast = getattr(nd, name)
            for x in (A, A.T, A[::2, 1::3], A[:0], nd.asarray(base[0, 0].reshape(()))):
                got, served, passed = _served(fast, x)
                assert (served, passed) == (1, 0), name
                _same(got, fast.__wrapped__(x))
                exp = getattr(np, name)(x.get())
                assert got.dtype == exp.dtype and got.shape == exp.shape


def test_calls_the_c_route_hands_to_python(eager):
    f = nd.asarray(np.arange(6, dtype=np.float32).reshape(2, 3))
    d = nd.asarray(np.arange(6, dtype=np.float64).reshape(2, 3))
    i = nd.asarray(np.arange(6, dtype=np.int32).reshape(2, 3))
    h = nd.asarray(np.arange(6, dtype=np.float16).reshape(2, 3))
    big = nd.asarray(np.zeros((300, 300), dtype=np.float32))
    cases = [
        (nd.add, (f, d)),                    # mixed float widths: NumPy's promotion
        (nd.add, (i, i)), (nd.add, (f, i)),  # integer operands
        (nd.add, (f, True)),                 # bool is not a weak int
        (nd.add, (f, np.float32(2))),        # a NumPy scalar is strong
        (nd.add, (f, 1 << 70)),              # beyond int64: travels as a double
        (nd.add, (h, h)),                    # storage-only dtype
        (nd.add, (f, [1.0, 2.0, 3.0])),      # a list
        (nd.add, (big, big.T)),              # a large transposed operand takes the tiled copy first
        (nd.sin, (i,)), (nd.sin, ([0.5, 1.0],)),
    ]
    with np.errstate(all="ignore"):
        for fn, args in cases:
            got, served, passed = _served(fn, *args)
            assert (served, passed) == (0, 1), (fn, args)
            host = [a.get() if isinstance(a, nd.DeviceArray) else a for a in args]
            exp = getattr(np, fn.__name__)(*host)
            assert got.dtype == exp.dtype
            np.testing.assert_allclose(got.get().astype(np.float64), np.asarray(exp, dtype=np.float64), rtol=1e-3)
    # keyword arguments go to Python whatever they are
    _, served, passed = _served(lambda: nd.add(f, f, dtype=None))
    assert (served, passed) == (0, 1)
    # errors are Python's: NumPy's message for a shape mismatch, TypeError for None
    with pytest.raises(ValueError, match="could not be broadcast"):
        nd.add(f, nd.asarray(np.zeros(4, dtype=np.float32)))
    with pytest.raises(TypeError):
        nd.add(f, None)
    with pytest.raises(TypeError):
        nd.add(f)


def test_matmul_where_reduce_and_bool_partner_routes(eager):
    rng = np.random.default_rng(9)
    for dtype in (np.float32, np.float64):
        a = rng.integers(-3, 4, (24, 16)).astype(dtype)
        b = rng.integers(-3, 4, (16, 40)).astype(dtype)
        A, B = nd.asarray(a), nd.asarray(b)
        At, Bt = nd.asarray(np.ascontiguousarray(a.T)), nd.asarray(np.ascontiguousarray(b.T))
        for x, y in ((A, B), (At.T, B), (A, Bt.T), (At.T, Bt.T), (A[::2], B[:, ::2])):
            got, served, passed = _served(nd.matmul, x, y)
            assert (served, passed) == (1, 0)
            _same(got, nd.matmul.__wrapped__(x, y))
            assert np.array_equal(got.get(), x.get().astype(np.float64) @ y.get())
        # where: bool condition, array / scalar branches, broadcasting
        c = nd.asarray(rng.random((24, 16)) < 0.5)
        row = nd.asarray(rng.standard_normal((16,)).astype(dtype))
        for x, y in ((A, 0), (0.5, A), (A, nd.multiply(A, 2.0)), (A, row), (row, A)):
            got, served, passed = _served(nd.where, c, x, y)
            assert (served, passed) == (1, 0)
            _same(got, nd.where.__wrapped__(c, x, y))
            hx = x.get() if isinstance(x, nd.DeviceArray) else x
            hy = y.get() if isinstance(y, nd.DeviceArray) else y
            exp = np.where(c.get(), hx, hy)
            assert got.dtype == exp.dtype and np.array_equal(got.get(), exp)
        got, served, _ = _served(nd.where, nd.asarray(np.array([True, False] * 8)), A, 1)     # a condition row broadcast over A
        assert served == 1 and np.array_equal(got.get(), np.where(np.array([True, False] * 8), a, 1))
        # a bool array next to a float one: the float loop
        m = nd.asarray(rng.random((24, 16)) < 0.5)
        for name in ("multiply", "add", "subtract", "true_divide", "maximum", "greater", "equal"):
            fn = getattr(nd, name)
            with np.errstate(all="ignore"):
                for x, y in ((A, m), (m, A)):
                    got, served, passed = _served(fn, x, y)
                    assert (served, passed) == (1, 0), name
                    _same(got, fn.__wrapped__(x, y))
                    exp = getattr(np, name)(x.get(), y.get())
                    assert got.dtype == exp.dtype
        # reductions of small arrays
        t = nd.asarray(rng.integers(-3, 4, (6, 5, 8)).astype(dtype))
        for name in ("sum", "prod", "max", "min"):
            fn = getattr(nd, name)
            forms = [((t,), {}), ((t,), {"axis": None}), ((t,), {"axis": 1}), ((t, 2), {}), ((t,), {"axis": (0, 2), "keepdims": True}),
                     ((t,), {"axis": -1, "keepdims": False, "dtype": None}), ((t,), {"axis": (2, 0, 1)}), ((t[:, ::2],), {"axis": (1,)}),
                     ((nd.transpose(t),), {"axis": 0})]
            for args, kw in forms:
                if "dtype" in kw and name in ("max", "min"):
                    continue                                 # (np.max takes no dtype)
                before = fp.stats()
                got = fn(*args, **kw)
                after = fp.stats()
                assert after["served"] - before["served"] == 1, (name, kw)
                _same(got, fn.__wrapped__(*args, **kw))
                hargs = [x.get() if isinstance(x, nd.DeviceArray) else x for x in args]
                exp = getattr(np, name)(*hargs, **kw)
                assert got.shape == exp.shape and got.dtype == exp.dtype
                np.testing.assert_allclose(got.get(), exp, rtol=1e-5)
        assert t.sum(axis=1).shape == (6, 8)            # the method spelling passes dtype=None / keepdims by keyword


def test_forms_the_new_routes_leave_to_python(eager):
    f = nd.asarray(np.arange(24, dtype=np.float32).reshape(4, 6))
    g = nd.asarray(np.arange(24, dtype=np.float64).reshape(6, 4))
    i = nd.asarray(np.arange(24, dtype=np.int32).reshape(6, 4))
    v = nd.asarray(np.arange(6, dtype=np.float32))
    b = nd.asarray(np.ones((4, 6), dtype=np.bool_))
    passed_cases = [
        (nd.matmul, (f, v), {}), (nd.matmul, (f, g), {}), (nd.matmul, (f, i), {}),            # a vector, mixed widths, integers
        (nd.matmul, (nd.asarray(np.ones((2, 4, 6), dtype=np.float32)), nd.asarray(np.ones((6, 3), dtype=np.float32))), {}),   # batched
        (nd.where, (b, 1, 2), {}), (nd.where, (b, f, np.float32(1)), {}), (nd.where, (f, f, 0), {}),   # two scalars, a NumPy scalar, a float condition
        (nd.sum, (f,), {"dtype": np.float64}), (nd.sum, (i,), {}), (nd.sum, (f,), {"axis": np.int64(0)}),
        (nd.sum, (nd.asarray(np.ones((1 << 9, 1 << 9), dtype=np.float32)),), {}),             # 2^18 elements: Python's staging rules apply
        (nd.multiply, (b, b), {}), (nd.multiply, (b, 2.0), {}),
    ]
    for fn, args, kw in passed_cases:
        before = fp.stats()
        got = fn(*args, **kw)
        after = fp.stats()
        assert after["served"] == before["served"] and after["passed"] - before["passed"] >= 1, (fn, kw)
        hargs = [x.get() if isinstance(x, nd.DeviceArray) else x for x in args]
        exp = getattr(np, fn.__name__)(*hargs, **kw)
        assert got.dtype == exp.dtype and got.shape == exp.shape
        np.testing.assert_allclose(got.get(), exp, rtol=1e-6)
    with pytest.raises(ValueError, match="mismatch in its core dimension"):
        nd.matmul(f, f)
    with pytest.raises(ValueError, match="duplicate"):
        nd.sum(f, axis=(0, 0))
    with pytest.raises(np.exceptions.AxisError):
        nd.sum(f, axis=2)
    with pytest.raises(ValueError, match="could not be broadcast"):
        nd.where(b, f, v[:5])


def test_lazy_mode_bypasses_the_c_route(lib):
    a = nd.asarray(np.arange(8, dtype=np.float32))
    prev = nd.set_lazy(True)
    try:
        assert fp.stats()["lazy"] == 1
        out, served, passed = _served(nd.multiply, a, a)
        assert (served, passed) == (0, 1) and out._expr is not None          # a pending expression, as before
        pending = out
    finally:
        nd.set_lazy(prev)
    nd.set_lazy(False)
    try:
        # eager call on a pending operand: Python materialises it
        out, served, passed = _served(nd.add, pending, a)
        assert (served, passed) == (0, 1)
        np.testing.assert_array_equal(out.get(), np.arange(8, dtype=np.float32) ** 2 + np.arange(8, dtype=np.float32))
        out, served, _ = _served(nd.add, pending, a)                          # .. after which it is an ordinary array
        assert served == 1
    finally:
        nd.set_lazy(prev)


def test_deferred_fill_is_run_by_python_before_the_c_route_reads_the_block(lib):
    prev = nd.set_lazy(True)
    try:
        rng = np.random.default_rng(0)
        x = nd.asarray(rng.standard_normal((64, 32)).astype(np.float32))
        cols = nd.sum(nd.multiply(x, x), axis=0)       # lazy mode may owe this column sum to a later pass
    finally:
        nd.set_lazy(False)
    try:
        out = nd.add(cols, 1.0)
        np.testing.assert_allclose(out.get(), (x.get() ** 2).sum(axis=0) + 1.0, rtol=1e-5)
    finally:
        nd.set_lazy(prev)


def test_buffer_and_array_objects(lib):
    assert nd._Buffer is fp.Buffer and issubclass(nd.DeviceArray, fp.ArrayBase)
    stats = (C.c_int64 * 4)()
    a = nd.DeviceArray.empty((1000,), np.float32)
    buf = a._buf
    assert type(buf) is fp.Buffer and buf.nbytes == 4000 and buf.ptr and buf.deps is None and buf.task is None
    buf.deps = {1: 2}
    assert buf.deps == {1: 2}
    buf.deps = None
    with pytest.raises(AttributeError):
        buf.ptr = 0
    r = weakref.ref(buf)
    ra = weakref.ref(a)
    lib.mem_stats(stats)
    in_use = stats[0]
    v = a[10:20]
    assert v._buf is buf and v._offset == 10 and v.shape == (10,) and v._code == a._code
    del a, buf
    gc.collect()
    assert r() is not None and ra() is None          # the view keeps the block
    del v
    gc.collect()
    assert r() is None
    lib.mem_stats(stats)
    assert stats[0] < in_use                          # .. and the block went back to the allocator with the last view
    # the constructor form the Python code uses, dtype code looked up when not given
    b = fp.Buffer(64)
    arr = nd.DeviceArray(b, 0, (4, 4), (4, 1), np.dtype(np.float32))
    assert arr._code == nd.dtype_code(np.float32) and arr._expr is None and arr._cdesc is None and arr._tasks is None
    with pytest.raises(TypeError):
        nd.DeviceArray(b, 0, [4, 4], (4, 1), np.dtype(np.float32))
    with pytest.raises(TypeError):
        nd.DeviceArray(b, 0, (4, 4), (4, 1), np.dtype(np.complex64))


def test_no_block_or_object_leak_over_many_calls(eager, lib):
    a = nd.asarray(np.ones((32, 8), dtype=np.float32))
    stats = (C.c_int64 * 4)()
    for _ in range(100):
        nd.add(nd.multiply(a, 2.0), nd.sin(a))
    gc.collect()
    lib.mem_stats(stats)
    in_use, n_obj = stats[0], len(gc.get_objects())
    for _ in range(5000):
        nd.add(nd.multiply(a, 2.0), nd.sin(a))
    with pytest.raises(ValueError):
        nd.add(a, nd.asarray(np.ones(5, dtype=np.float32)))
    gc.collect()
    lib.mem_stats(stats)
    assert stats[0] == in_use
    assert len(gc.get_objects()) - n_obj < 50
    # reference counts of the shared tuples / dtype objects are balanced: the operand is still intact
    assert a.shape == (32, 8) and a._strides == (8, 1) and a.dtype == np.float32


def test_tape_results_identical_with_and_without_the_c_route(engines, eager):
    hip, _ = engines
    rng = np.random.default_rng(5)
    xv, yv = rng.standard_normal((2, 4)).astype(np.float32), rng.standard_normal((2, 4)).astype(np.float32)

    def sweep():
        x, y = hip.Tensor(xv, allow_grad=True), hip.Tensor(yv, allow_grad=True)
        f = 2 * y * hip.sin(x) - x ** 2
        f.backward(allow_higher_order=True)
        x.grad.backward()
        return x.grad.as_numpy().copy(), y.grad.as_numpy().copy()

    before = fp.stats()["served"]
    on = sweep()
    assert fp.stats()["served"] - before >= 20
    prev = fp.enable_ops(False)
    try:
        before = fp.stats()["served"]
        off = sweep()
        assert fp.stats()["served"] == before
    finally:
        fp.enable_ops(prev)
    for g, e in zip(on, off):
        np.testing.assert_array_equal(g, e)


@pytest.mark.gpu
def test_c_route_on_the_device_gpu(on_gpu, eager):
    assert on_gpu
    rng = np.random.default_rng(6)
    a = rng.standard_normal((513, 257)).astype(np.float32)
    b = rng.standard_normal((257,)).astype(np.float32)
    A, B = nd.asarray(a), nd.asarray(b)
    out, served, passed = _served(nd.add, nd.multiply(A, B), nd.sin(A))
    assert served == 1
    np.testing.assert_allclose(out.get(), a * b + np.sin(a), rtol=2e-6, atol=2e-6)
    _same(nd.multiply(A, 2.0), nd.multiply.__wrapped__(A, 2.0))
    _same(nd.greater(A, 0), nd.greater.__wrapped__(A, 0))
