"""Shim-vs-NumPy cases shared by the CPU (test double) and GPU (product) runs.

Each case builds seeded inputs, calls minidiff_amd.ndarray (which goes through
the C-ABI) and numpy on the same data, and compares: dtype and shape exactly,
integer / bool / index results bit-for-bit, floats within the stated bound
(north_star: 1e-5 relative for fp32; we hold 2e-6 here, 1e-13 for fp64).
"""
import numpy as np

F32_RTOL, F64_RTOL = 2e-6, 1e-13


def check(name, got, exp, exact=False, rtol=None):
    got = np.asarray(got.get() if hasattr(got, "get") else got)
    exp = np.asarray(exp)
    assert got.shape == exp.shape, (name, got.shape, exp.shape)
    assert got.dtype == exp.dtype, (name, got.dtype, exp.dtype)
    if exact or exp.dtype.kind in "biu":
        assert np.array_equal(got, exp, equal_nan=exp.dtype.kind == "f"), (name, got, exp)
        return
    if rtol is None:
        rtol = F32_RTOL if exp.dtype == np.float32 else F64_RTOL
    scale = float(np.nanmax(np.abs(exp[np.isfinite(exp)]))) if np.isfinite(exp).any() else 1.0
    ok = np.allclose(got, exp, rtol=rtol, atol=rtol * max(scale, 1e-30), equal_nan=True)
    assert ok, (name, float(np.nanmax(np.abs(got - exp))), scale)


def _data(dtype, shape=(2, 3, 4), seed=0):
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "f":
        return rng.standard_normal(shape).astype(dtype)
    if np.dtype(dtype).kind == "b":
        return rng.integers(0, 2, shape).astype(bool)
    return rng.integers(-6, 7, shape).astype(dtype)


BINARY = ["add", "subtract", "multiply", "true_divide", "power", "mod", "floor_divide", "maximum", "minimum", "less",
          "less_equal", "greater", "greater_equal", "equal", "not_equal", "logical_and", "logical_or", "logical_xor"]
UNARY_F = ["absolute", "sign", "ceil", "floor", "sin", "cos", "tan", "sinh", "cosh", "tanh", "exp", "log", "sqrt",
           "logical_not", "negative"]
REDUCE = ["sum", "prod", "max", "min", "mean", "std", "any", "all"]


def case_binary(nd, name, dtype):
    x, y = _data(dtype, (2, 3, 4), 1), _data(dtype, (3, 1), 2)
    if name == "power":
        x = np.abs(x) + (1 if np.dtype(dtype).kind == "i" else 0.5)
        x = x.astype(dtype)
        if np.dtype(dtype).kind == "i":
            y = np.abs(y)
    dx, dy = nd.asarray(x), nd.asarray(y)
    f, g = getattr(nd, name), getattr(np, name)
    with np.errstate(all="ignore"):
        check(name, f(dx, dy), g(x, y))
        check(name + "/py-int", f(dx, 2), g(x, 2))
        check(name + "/py-float-left", f(2.5, dx), g(2.5, x))
        check(name + "/np-scalar", f(dx, np.float32(1.5)), g(x, np.float32(1.5)))
        check(name + "/view", f(dx.T, dx.T), g(x.T, x.T))
        check(name + "/0d", f(dx, nd.asarray(np.array(3, dtype=dtype))), g(x, np.array(3, dtype=dtype)))


def case_binary_mixed(nd, name):
    xf, xi, xb = _data(np.float32, (4, 5), 3), _data(np.int64, (4, 5), 4), _data(np.bool_, (4, 5), 5)
    f, g = getattr(nd, name), getattr(np, name)
    with np.errstate(all="ignore"):
        for a, b in ((xf, xi), (xi, xf), (xf, xb), (xi, xb), (xi.astype(np.int32), xi), (xf, xf.astype(np.float64))):
            if name in ("power",) and (b.dtype.kind == "i"):
                b = np.abs(b)
            if name == "subtract" and a.dtype == b.dtype == np.bool_:
                continue
            check(f"{name}/{a.dtype}-{b.dtype}", f(nd.asarray(a), nd.asarray(b)), g(a, b))


def case_unary(nd, name, dtype):
    x = _data(dtype, (3, 5, 2), 6)
    if name in ("log", "sqrt"):
        x = np.abs(x) + np.asarray(0.25, dtype=dtype)
    dx = nd.asarray(x)
    with np.errstate(all="ignore"):
        check(name, getattr(nd, name)(dx), getattr(np, name)(x))
        check(name + "/view", getattr(nd, name)(dx.T[::2]), getattr(np, name)(x.T[::2]))


def case_unary_int(nd):
    xi = _data(np.int64, (4, 5), 7)
    dxi = nd.asarray(xi)
    for name in ("absolute", "sign", "negative", "invert", "sin", "exp", "ceil", "floor", "logical_not"):
        with np.errstate(all="ignore"):
            check(name + "/i64", getattr(nd, name)(dxi), getattr(np, name)(xi))
    xb = _data(np.bool_, (4, 5), 8)
    check("invert/bool", nd.invert(nd.asarray(xb)), np.invert(xb))
    check("logical_not/bool", nd.logical_not(nd.asarray(xb)), np.logical_not(xb))
    check("pow/int", nd.power(dxi, 3), np.power(xi, 3))
    check("pow/int2", nd.power(dxi, 2), np.power(xi, 2))
    try:
        nd.power(dxi, -1)
        raise AssertionError("negative integer power must raise")
    except ValueError:
        pass


def case_reduce(nd, name, dtype):
    x = _data(dtype, (3, 4, 5), 9)
    dx = nd.asarray(x)
    for ax in (None, 0, 1, 2, (0, 2), (1, 2), (0, 1, 2), -1, ()):
        for kd in (False, True):
            with np.errstate(all="ignore"):
                check(f"{name}{ax}{kd}", getattr(nd, name)(dx, axis=ax, keepdims=kd), getattr(np, name)(x, axis=ax, keepdims=kd),
                      rtol=1e-5 if np.dtype(dtype) == np.float32 else None)
    check(name + "/T", getattr(nd, name)(nd.transpose(dx), axis=(0,)), getattr(np, name)(x.T, axis=(0,)),
          rtol=1e-5 if np.dtype(dtype) == np.float32 else None)


def case_reduce_large(nd):
    """Shapes that reach the rows / cols / split code paths of the device."""
    rng = np.random.default_rng(10)
    x = rng.standard_normal((1030, 517)).astype(np.float32)
    dx = nd.asarray(x)
    check("sum/all", nd.sum(dx), np.sum(x), rtol=1e-5)
    check("sum/cols", nd.sum(dx, axis=0), np.sum(x, axis=0), rtol=1e-5)
    check("sum/rows", nd.sum(dx, axis=1), np.sum(x, axis=1), rtol=1e-5)
    check("max/cols", nd.max(dx, axis=0), np.max(x, axis=0))
    check("min/rows", nd.min(dx, axis=1), np.min(x, axis=1))
    check("mean/all", nd.mean(dx), np.mean(x), rtol=1e-5)
    check("argmax/rows", nd.argmax(dx, axis=1), np.argmax(x, axis=1))
    check("argmin/flat", nd.argmin(dx), np.argmin(x))
    check("sum/T-all", nd.sum(dx.T), np.sum(x.T), rtol=1e-5)
    w = rng.standard_normal((1500, 768)).astype(np.float32)  # vectorised column reduce (+ split + finishing pass)
    dw = nd.asarray(w)
    check("sum/cols-vec", nd.sum(dw, axis=0), np.sum(w, axis=0), rtol=1e-5)
    check("max/cols-vec", nd.max(dw, axis=0), np.max(w, axis=0))
    check("sum/cols-vec-view", nd.sum(dw[4:, 8:520], axis=(0,)), np.sum(w[4:, 8:520], axis=(0,)), rtol=1e-5)
    w64 = rng.standard_normal((515, 256))
    check("sum/cols-vec-f64", nd.sum(nd.asarray(w64), axis=0), np.sum(w64, axis=0), rtol=1e-12)
    check("sum/cols-short", nd.sum(dw[:17], axis=0), np.sum(w[:17], axis=0), rtol=1e-5)
    y = rng.standard_normal((300_001,)).astype(np.float32)
    check("sum/1d-odd", nd.sum(nd.asarray(y)), np.sum(y), rtol=1e-5)
    check("sum/1d-offset", nd.sum(nd.asarray(y)[3:]), np.sum(y[3:]), rtol=1e-5)
    xi = rng.integers(-1000, 1000, (2049, 131)).astype(np.int64)
    dxi = nd.asarray(xi)
    check("isum/all", nd.sum(dxi), np.sum(xi))
    check("isum/cols", nd.sum(dxi, axis=0), np.sum(xi, axis=0))
    check("isum/rows", nd.sum(dxi, axis=1), np.sum(xi, axis=1))
    z = rng.standard_normal((8, 300, 40))
    check("sum/mid", nd.sum(nd.asarray(z), axis=1), np.sum(z, axis=1), rtol=1e-12)
    check("any", nd.any(dxi > 998), np.any(xi > 998))
    check("all", nd.all(dxi > -1000, axis=0), np.all(xi > -1000, axis=0))


def case_argreduce(nd):
    x = _data(np.float64, (4, 5, 6), 11)
    x[1, 2, 3] = np.nan
    dx = nd.asarray(x)
    for ax in (None, 0, 1, 2, -1):
        check(f"argmax{ax}", nd.argmax(dx, axis=ax), np.argmax(x, axis=ax))
        check(f"argmin{ax}k", nd.argmin(dx, axis=ax, keepdims=True), np.argmin(x, axis=ax, keepdims=True))
    xi = np.array([[1, 5, 5, 2], [7, 7, 0, 0]])
    check("argmax/ties", nd.argmax(nd.asarray(xi), axis=1), np.argmax(xi, axis=1))
    for bad in ((0, 1), [0]):
        try:
            nd.argmax(dx, axis=bad)
            raise AssertionError("tuple axis must raise TypeError like numpy")
        except TypeError:
            pass


def case_layout(nd):
    x, y = _data(np.float32, (2, 3, 4), 12), _data(np.float32, (3, 1), 13)
    dx, dy = nd.asarray(x), nd.asarray(y)
    check("T", dx.T, x.T)
    check("transpose", nd.transpose(dx, (1, 0, 2)), x.transpose(1, 0, 2))
    check("reshape/copy", nd.reshape(dx.T, (6, 4)), x.T.reshape(6, 4))
    check("reshape/view", nd.reshape(dx, (6, 4)), x.reshape(6, 4))
    check("reshape/-1", nd.reshape(dx, (-1, 2)), x.reshape(-1, 2))
    check("flip", nd.flip(dx, (0, 2)), np.flip(x, (0, 2)))
    check("flip/all", nd.flip(dx), np.flip(x))
    check("flip+op", nd.add(nd.flip(dx, 1), 1), np.flip(x, 1) + 1)
    check("broadcast_to", nd.broadcast_to(dy, (2, 3, 4)), np.broadcast_to(y, (2, 3, 4)))
    check("expand_dims", nd.expand_dims(dx, (0, 2)), np.expand_dims(x, (0, 2)))
    check("squeeze", nd.squeeze(nd.expand_dims(dx, 1)), x)
    check("atleast_3d", nd.atleast_3d(dy), np.atleast_3d(y))
    check("ravel", nd.ravel(dx.T), x.T.ravel())
    check("flatten", nd.flatten(dx), x.flatten())
    check("swapaxes", nd.swapaxes(dx, 0, 2), np.swapaxes(x, 0, 2))
    check("tile", nd.tile(dx, (2, 1, 3)), np.tile(x, (2, 1, 3)))
    check("tile/more", nd.tile(dy, (2, 2, 2, 2)), np.tile(y, (2, 2, 2, 2)))
    check("repeat", nd.repeat(dx, 3, axis=1), np.repeat(x, 3, axis=1))
    check("repeat/arr", nd.repeat(dx, [1, 0, 2], axis=1), np.repeat(x, [1, 0, 2], axis=1))
    check("concatenate", nd.concatenate([dx, dx], axis=1), np.concatenate([x, x], axis=1))
    check("stack", nd.stack([dx, dx], axis=2), np.stack([x, x], axis=2))
    for p, q in zip(nd.split(dx, 2, axis=2), np.split(x, 2, axis=2)):
        check("split", p, q)
    check("arange", nd.arange(5), np.arange(5))
    check("arange/f", nd.arange(1, 2, 0.25), np.arange(1, 2, 0.25))
    check("astype/i64", dx.astype(np.int64), x.astype(np.int64))
    check("astype/bool", dx.astype(bool), x.astype(bool))
    check("astype/f64", dx.astype(np.float64), x.astype(np.float64))
    check("full", nd.full((2, 2), 3), np.full((2, 2), 3))
    check("ones", nd.ones((2,)), np.ones((2,)))
    check("zeros_like", nd.zeros_like(dx), np.zeros_like(x))
    check("copy/strided", nd.copy(dx[:, ::2, 1:]), x[:, ::2, 1:].copy())


def case_matmul(nd, dtype):
    rng = np.random.default_rng(14)
    A = rng.standard_normal((10, 30)).astype(dtype)
    B = rng.standard_normal((30, 20)).astype(dtype)
    dA, dB = nd.asarray(A), nd.asarray(B)
    rt = 2e-6 if np.dtype(dtype) == np.float32 else 1e-13
    check("mm", nd.matmul(dA, dB), A @ B, rtol=rt)
    check("mm/TT", nd.matmul(dB.T, dA.T), B.T @ A.T, rtol=rt)
    check("mv", nd.matmul(dA, dB[:, 0]), A @ B[:, 0], rtol=rt)
    check("vm", nd.matmul(dA[0], dB), A[0] @ B, rtol=rt)
    A3 = rng.standard_normal((2, 3, 4, 5)).astype(dtype)
    B3 = rng.standard_normal((3, 5, 6)).astype(dtype)
    check("bmm", nd.matmul(nd.asarray(A3), nd.asarray(B3)), A3 @ B3, rtol=rt)
    check("dot/1d", nd.dot(dA[0], dA[1]), np.dot(A[0], A[1]), rtol=rt)
    T1 = rng.standard_normal((2, 2, 2, 2)).astype(dtype)
    check("tensordot", nd.tensordot(nd.asarray(T1), nd.asarray(T1)), np.tensordot(T1, T1), rtol=rt)
    check("tensordot/axes", nd.tensordot(nd.asarray(T1), nd.asarray(T1), axes=([0, 2], [1, 3])),
          np.tensordot(T1, T1, axes=([0, 2], [1, 3])), rtol=rt)
    try:
        nd.matmul(dA, dA)
        raise AssertionError("shape mismatch must raise ValueError")
    except ValueError:
        pass


def case_matmul_mfma(nd):
    """Sizes that take the MFMA kernel: all four operand layouts, full and ragged tiles."""
    rng = np.random.default_rng(15)
    for (M, K, N) in ((256, 128, 384), (200, 72, 136), (128, 16, 128), (130, 33, 257)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((K, N)).astype(np.float32)
        At, Bt = np.ascontiguousarray(A.T), np.ascontiguousarray(B.T)
        dA, dB, dAt, dBt = nd.asarray(A), nd.asarray(B), nd.asarray(At), nd.asarray(Bt)
        exp = A.astype(np.float64) @ B.astype(np.float64)
        for tag, a, b in (("NN", dA, dB), ("NT", dA, dBt.T), ("TN", dAt.T, dB), ("TT", dAt.T, dBt.T)):
            got = nd.matmul(a, b).get()
            assert got.dtype == np.float32 and got.shape == (M, N)
            err = np.abs(got - exp).max() / np.abs(exp).max()
            assert err < 2e-6, (tag, M, K, N, err)
    # asymmetric B with A = I catches a transposed C write
    n = 128
    Bm = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 97) - 40
    got = nd.matmul(nd.asarray(np.eye(n, dtype=np.float32)), nd.asarray(Bm)).get()
    assert np.array_equal(got, Bm)
    # batched through the MFMA kernel
    A3 = rng.standard_normal((3, 128, 64)).astype(np.float32)
    B3 = rng.standard_normal((3, 64, 128)).astype(np.float32)
    check("bmm/mfma", nd.matmul(nd.asarray(A3), nd.asarray(B3)), A3 @ B3, rtol=2e-6)


def case_where_clip(nd):
    x, y = _data(np.float32, (2, 3, 4), 16), _data(np.float32, (3, 1), 17)
    dx, dy = nd.asarray(x), nd.asarray(y)
    c = x > 0
    check("where/scalar", nd.where(nd.asarray(c), dx, 0), np.where(c, x, 0))
    check("where/bcast", nd.where(dx > 0, dx, dy), np.where(x > 0, x, y))
    check("where/int-cond", nd.where(nd.asarray(c.astype(np.int64)), 1.5, dx), np.where(c.astype(np.int64), 1.5, x))
    check("clip", nd.clip(dx, -0.5, 0.5), np.clip(x, -0.5, 0.5))
    check("clip/min", nd.clip(dx, 0, None), np.clip(x, 0, None))
    check("mask-mul", nd.multiply(dx, nd.asarray(c)), x * c)
    check("mask-mul/left", nd.multiply(nd.asarray(c), dx), c * x)
    big = _data(np.float32, (64, 256), 18)
    db = nd.asarray(big)
    bias = _data(np.float32, (256,), 19)
    check("bias-add", nd.add(db, nd.asarray(bias)), big + bias)
    check("col-bcast", nd.multiply(db, nd.asarray(bias[:64, None])), big * bias[:64, None])
    z = big + bias
    check("relu", nd.where(nd.greater(nd.asarray(z), 0), nd.asarray(z), 0), np.where(z > 0, z, 0))
    seed = nd.broadcast_to(nd.asarray(np.float32(1.0)), big.shape)
    check("seed*mask", nd.multiply(seed, nd.asarray(z > 0)), np.broadcast_to(np.float32(1.0), big.shape) * (z > 0))
    check("seed*2", nd.multiply(seed, 2), np.broadcast_to(np.float32(1.0), big.shape) * 2)


def case_indexing(nd):
    x = _data(np.float32, (2, 3, 4), 20)
    dx = nd.asarray(x)
    c = x > 0
    check("int", dx[1], x[1])
    check("slices", dx[:, 1:3, ::2], x[:, 1:3, ::2])
    check("ellipsis", dx[..., None, -1], x[..., None, -1])
    check("neg-step", dx[::-1], x[::-1])
    idx = np.array([1, 0, 1, 1])
    check("adv/axis0", dx[nd.asarray(idx)], x[idx])
    check("adv/mid", dx[:, idx[:3], :], x[:, idx[:3], :])
    check("adv/split", dx[idx[:2], :, idx[:2]], x[idx[:2], :, idx[:2]])
    check("adv/int+arr", dx[0, :, idx], x[0, :, idx])
    check("adv/lists", dx[:, [0, 2], [1, 3]], x[:, [0, 2], [1, 3]])
    check("adv/bool", dx[nd.asarray(c)], x[c])
    check("adv/neg", dx[nd.asarray(np.array([-1, -2]))], x[np.array([-1, -2])])
    i2 = (np.arange(2)[:, None], np.array([[0, 2]]), np.array([[3, 1]]))
    check("adv/bcast", dx[tuple(nd.asarray(i) for i in i2)], x[i2])
    try:
        dx[nd.asarray(np.array([5]))]
        raise AssertionError("out of range index must raise IndexError")
    except IndexError:
        pass
    z2, dz = x.copy(), nd.asarray(x.copy())
    np.add.at(z2, idx, 1.0)
    nd.index_add(dz, nd.asarray(idx), 1.0)
    check("add.at/scalar", dz, z2, exact=True)
    v = _data(np.float32, (4, 3, 4), 21)
    z3, dz3 = x.copy(), nd.asarray(x.copy())
    np.add.at(z3, idx, v)
    nd.index_add(dz3, nd.asarray(idx), nd.asarray(v))
    check("add.at/dups", dz3, z3, exact=True)
    zi, dzi = _data(np.int64, (5, 3), 22), None
    dzi = nd.asarray(zi.copy())
    ii = np.array([4, 4, 0, 4])
    np.add.at(zi, ii, 3)
    nd.index_add(dzi, nd.asarray(ii), 3)
    check("add.at/int", dzi, zi)
    z4, dz4 = x.copy(), nd.asarray(x.copy())
    z4[:, 1] = 7
    dz4[:, 1] = 7
    check("set/basic", dz4, z4)
    z4[idx[:2], 0, idx[:2]] = [1, 2]
    dz4[idx[:2], 0, idx[:2]] = [1, 2]
    check("set/adv", dz4, z4)
    z4[idx] = 9.0  # duplicates, same value
    dz4[nd.asarray(idx)] = 9.0
    check("set/dups", dz4, z4)
    am = np.argmax(x, axis=1, keepdims=True)
    check("take_along_axis", nd.take_along_axis(dx, nd.asarray(am), 1), np.take_along_axis(x, am, 1))
    z5, dz5 = np.zeros_like(x), nd.zeros_like(dx)
    np.put_along_axis(z5, am, 5.0, 1)
    nd.put_along_axis(dz5, nd.asarray(am), 5.0, 1)
    check("put_along_axis", dz5, z5)
    # large plans reach the ordered-rounds scatter
    rng = np.random.default_rng(23)
    big = rng.standard_normal((20000,)).astype(np.float32)
    keys = rng.integers(0, 500, (20000,))
    acc, dacc = np.zeros(500, np.float32), nd.zeros((500,), dtype=np.float32)
    np.add.at(acc, keys, big)
    nd.index_add(dacc, nd.asarray(keys), nd.asarray(big))
    check("add.at/large-dups", dacc, acc, exact=True)
    dst, ddst = np.zeros(500, np.float32), nd.zeros((500,), dtype=np.float32)
    dst[keys] = big
    ddst[nd.asarray(keys)] = nd.asarray(big)
    check("set/large-dups", ddst, dst, exact=True)
    check("gather/large", nd.asarray(big)[nd.asarray(keys)], big[keys], exact=True)


def case_index_utils(nd):
    rng = np.random.default_rng(30)
    m = rng.integers(0, 3, (5, 7, 3)) > 1
    dm = nd.asarray(m)
    for got, exp in zip(nd.nonzero(dm), np.nonzero(m)):
        check("nonzero", got, exp)
    check("argwhere", nd.argwhere(dm), np.argwhere(m))
    check("argwhere/float", nd.argwhere(nd.asarray(m * 1.5)), np.argwhere(m * 1.5))
    check("argwhere/none", nd.argwhere(nd.asarray(np.zeros((3, 2)))), np.argwhere(np.zeros((3, 2))))
    big = rng.integers(0, 5, (70001,)) == 0          # several compaction blocks, ragged tail
    check("flatnonzero/big", nd.flatnonzero(nd.asarray(big)), np.flatnonzero(big))
    x = rng.standard_normal((5, 7, 3)).astype(np.float32)
    check("bool-mask getitem", nd.asarray(x)[dm], x[m], exact=True)
    check("bool-mask partial", nd.asarray(x)[nd.asarray(m[:, :, 0])], x[m[:, :, 0]], exact=True)
    xi = rng.integers(0, 10, (4, 6))
    check("isin", nd.isin(nd.asarray(xi), [1, 3, 7]), np.isin(xi, [1, 3, 7]))
    check("isin/arr", nd.isin(nd.asarray(xi), nd.asarray(np.array([2, 9]))), np.isin(xi, np.array([2, 9])))
    flat = rng.integers(0, 5 * 7 * 3, (11,))
    for got, exp in zip(nd.unravel_index(nd.asarray(flat), (5, 7, 3)), np.unravel_index(flat, (5, 7, 3))):
        check("unravel_index", got, exp)
    try:
        nd.unravel_index(nd.asarray(np.array([105])), (5, 7, 3))
        raise AssertionError("out-of-range flat index must raise ValueError")
    except ValueError:
        pass


def case_inplace(nd):
    x, y = _data(np.float32, (2, 3, 4), 24), _data(np.float32, (3, 1), 25)
    w, dw = x.copy(), nd.asarray(x.copy())
    w += y; dw += nd.asarray(y)
    check("iadd", dw, w)
    w *= 2; dw *= 2
    check("imul", dw, w)
    w **= 2; dw **= 2
    check("ipow", dw, w)
    w -= 1.5; dw -= 1.5
    check("isub", dw, w)
    w /= 3; dw /= 3
    check("itruediv", dw, w)
    v, dv = w[:, 1], dw[:, 1]
    v += 1; dv += 1
    check("iadd/view-aliases-base", dw, w)
    wi, dwi = _data(np.int64, (3, 3), 26), None
    dwi = nd.asarray(wi.copy())
    wi //= 2; dwi //= 2
    wi %= 3; dwi %= 3
    check("ifloordiv/imod", dwi, wi)
    try:
        dwi += 0.5
        raise AssertionError("int += float must raise (same_kind casting)")
    except TypeError:
        pass
    m, dm = _data(np.float32, (4, 4), 27), None
    dm = nd.asarray(m.copy())
    m @= m.copy()
    dm @= dm.copy()
    check("imatmul", dm, m, rtol=1e-5)


def case_errors(nd):
    x = nd.asarray(np.zeros((2, 3)))
    for fn, exc in ((lambda: nd.add(x, nd.asarray(np.zeros((4,)))), ValueError),
                    (lambda: nd.sum(x, axis=5), (ValueError, IndexError)),
                    (lambda: nd.reshape(x, (4, 2)), ValueError),
                    (lambda: nd.asarray(np.zeros(3, dtype=np.complex64)), TypeError),
                    (lambda: nd.subtract(nd.asarray(np.array([True])), nd.asarray(np.array([False]))), TypeError),
                    (lambda: nd.transpose(x, (0, 0)), ValueError),
                    (lambda: x[2], IndexError)):
        try:
            fn()
        except exc:
            continue
        raise AssertionError("expected " + str(exc))
    # empties
    e = nd.asarray(np.zeros((0, 3), dtype=np.float32))
    check("empty/add", nd.add(e, 1), np.zeros((0, 3), np.float32) + 1)
    check("empty/sum", nd.sum(e, axis=0), np.zeros((0, 3), np.float32).sum(axis=0))
    check("empty/sum-all", nd.sum(e), np.zeros((0, 3), np.float32).sum())
