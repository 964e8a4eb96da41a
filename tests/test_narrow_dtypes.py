"""Storage-only dtypes of the reference table (minidiff/backend/numpy.py:188-200: float16, int8/16, uint8/16/32/64) against
NumPy on seeded inputs: result dtypes, integer results bit-exact (wrap-around included), float16 within its precision; views,
transfers, conversions, gathers / scatters, in-place forms, creation functions. The reference-generated golden cases for
these dtypes are in tests/golden (narrow_*); csrc/md_narrow.h and minidiff_amd/narrow.py have the design."""
import numpy as np
import pytest

NARROW = (np.int8, np.int16, np.uint8, np.uint16, np.uint32, np.uint64, np.float16)
gpu = pytest.mark.gpu


def _run(nd, dt):
    rng = np.random.default_rng(int(np.dtype(dt).num))
    a = rng.integers(0, 100, (5, 7)).astype(dt)
    b = rng.integers(1, 50, (7,)).astype(dt)
    da, db = nd.asarray(a), nd.asarray(b)
    assert da.dtype == a.dtype and np.array_equal(da.get(), a)
    for name in ("add", "subtract", "multiply", "floor_divide", "mod", "true_divide", "maximum", "minimum", "less", "equal", "not_equal", "power"):
        with np.errstate(all="ignore"):
            ref = getattr(np, name)(a, b)
        got = getattr(nd, name)(da, db)
        assert got.dtype == ref.dtype, (dt, name, got.dtype, ref.dtype)
        g = got.get()
        ok = np.array_equal(g, ref) if ref.dtype.kind != "f" else np.allclose(g, ref, rtol=2e-3, equal_nan=True)
        assert ok, (dt, name)
    for name in ("negative", "absolute", "sign"):
        if np.dtype(dt).kind == "u" and name == "negative":
            continue
        ref = getattr(np, name)(a)
        got = getattr(nd, name)(da)
        assert got.dtype == ref.dtype and np.array_equal(got.get(), ref), (dt, name)
    for name in ("sum", "max", "min", "mean", "argmax"):
        with np.errstate(all="ignore"):
            ref = getattr(np, name)(a, axis=0)
        got = getattr(nd, name)(da, axis=0)
        assert got.dtype == ref.dtype, (dt, name, got.dtype, ref.dtype)
        assert np.allclose(got.get().astype(np.float64), ref.astype(np.float64), rtol=2e-3), (dt, name)
    # scalars: weak Python scalars keep the narrow type, as in NumPy (NEP 50)
    for s in (3, 2.5):
        if np.dtype(dt).kind in "iu" and isinstance(s, float):
            ref, got = a * s, nd.multiply(da, s)
        else:
            ref, got = a + s, nd.add(da, s)
        assert got.dtype == ref.dtype and np.allclose(got.get().astype(np.float64), ref.astype(np.float64), rtol=2e-3), (dt, s)
    # views and transfers: strides only
    assert np.array_equal(da.T.get(), a.T) and np.array_equal(da[1:4, ::2].get(), a[1:4, ::2])
    assert np.array_equal(nd.reshape(da, (7, 5)).get(), a.reshape(7, 5)) and np.array_equal(nd.flip(da, axis=1).get(), a[:, ::-1])
    # conversions, every direction
    for other in NARROW + (np.bool_, np.int32, np.int64, np.float32, np.float64):
        with np.errstate(all="ignore"):
            ref = a.astype(other)
        got = da.astype(other)
        assert got.dtype == ref.dtype and np.array_equal(got.get(), ref), (dt, other)
    # gather / scatter with duplicates, in-place arithmetic, np.add.at
    idx = nd.asarray(np.array([3, 1, 1, 0]))
    assert np.array_equal(da[idx].get(), a[[3, 1, 1, 0]])
    c = da.copy(); c[idx] = db; a2 = a.copy(); a2[[3, 1, 1, 0]] = b
    assert np.array_equal(c.get(), a2)
    c = da.copy(); c += db; a2 = a.copy(); a2 += b
    assert np.array_equal(c.get(), a2)
    c = da.copy(); nd.index_add(c, idx, db); a2 = a.copy(); np.add.at(a2, [3, 1, 1, 0], b)
    assert np.array_equal(c.get(), a2), dt
    if np.dtype(dt).kind in "iu":
        with pytest.raises(TypeError):      # NumPy's same_kind casting rule for in-place updates
            c = da.copy(); c += 0.5
    # creation and joining
    z = nd.zeros((3, 4), dtype=dt); o = nd.ones((3,), dtype=dt); f = nd.full((2, 2), 7, dtype=dt)
    assert z.dtype == dt and np.array_equal(o.get(), np.ones(3, dt)) and np.array_equal(f.get(), np.full((2, 2), 7, dt))
    assert np.array_equal(nd.concatenate([da, da], axis=1).get(), np.concatenate([a, a], axis=1))
    assert np.array_equal(nd.stack([da, da]).get(), np.stack([a, a]))
    assert np.array_equal(nd.where(nd.greater(da, 50), da, 0).get(), np.where(a > 50, a, 0))
    assert np.array_equal(nd.take_along_axis(da, nd.asarray(np.argsort(a, axis=1)), 1).get(), np.take_along_axis(a, np.argsort(a, axis=1), 1))


@pytest.mark.parametrize("dt", NARROW, ids=lambda d: np.dtype(d).name)
def test_narrow_dtype_cpu(lib, on_gpu, dt):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _run(nd, dt)


@gpu
@pytest.mark.parametrize("dt", NARROW, ids=lambda d: np.dtype(d).name)
def test_narrow_dtype_gpu(lib, on_gpu, dt):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _run(nd, dt)


def test_uint64_whole_range_in_the_native_path(lib):
    """Elementwise arithmetic, comparisons, where and the reductions have unsigned 64-bit loops of their own (csrc/md_narrow.h: the
    uint64 carrier): values >= 2**63 give NumPy's results; gathers / scatters move and add in the array's own type. The functions
    that still go promote -> wide kernel -> demote (narrow.COMPUTE: products, statistics, builders) give the same bits either way
    or convert each element from its own type first: no function of the table refuses such values any more."""
    from minidiff_amd import ndarray as nd
    big = np.array([1, 2 ** 63 + 5, 7, 2 ** 64 - 1, 2 ** 63], dtype=np.uint64)
    oth = np.array([3, 2 ** 63 + 1, 2 ** 64 - 1, 2, 5], dtype=np.uint64)
    d, o = nd.asarray(big), nd.asarray(oth)
    assert np.array_equal(d.get(), big) and np.array_equal(d[::-1].get(), big[::-1])       # storage, views: fine
    assert np.array_equal(nd.concatenate([d, d]).get(), np.concatenate([big, big]))         # movers: fine (bits)
    assert np.array_equal(d.astype(np.float64).get(), big.astype(np.float64))              # conversion kernel: true values
    for name in ("add", "subtract", "multiply", "floor_divide", "mod", "maximum", "minimum", "less", "greater_equal", "equal", "power"):
        with np.errstate(all="ignore"):
            ref = getattr(np, name)(big, oth)
        got = getattr(nd, name)(d, o)
        assert got.dtype == ref.dtype and np.array_equal(got.get(), ref), name
    assert np.array_equal(nd.add(d, 2 ** 63 + 9).get(), big + np.uint64(2 ** 63 + 9))      # a Python int beyond int64 in a uint64 loop
    with pytest.raises(OverflowError):
        nd.add(d, -1)                                                                       # NEP 50: must fit the loop dtype
    for name in ("max", "min", "sum", "prod", "argmax", "argmin", "any", "all"):
        with np.errstate(all="ignore"):
            ref = getattr(np, name)(big)
        got = getattr(nd, name)(d)
        assert got.dtype == ref.dtype and np.array_equal(np.asarray(got.get()), ref), name
    assert np.array_equal(nd.where(nd.greater(d, o), d, o).get(), np.where(big > oth, big, oth))
    # uint64 with int64: NumPy's loop is float64 — each operand converted first, no detour through int64
    i64 = np.array([2 ** 62, -1, 3, -(2 ** 62), 0], dtype=np.int64)
    ref = big + i64
    got = nd.add(d, nd.asarray(i64))
    assert got.dtype == ref.dtype == np.float64 and np.array_equal(got.get(), ref)
    for name in ("mean", "std"):                                                            # read as float64, NumPy's first step
        assert np.array_equal(getattr(nd, name)(d).get(), getattr(np, name)(big)), name
    # products: sums of products wrap mod 2**64 whichever way the bits are read; against a signed operand NumPy's loop is float64
    m, k = big.reshape(5, 1) * np.uint64(3) + oth, oth.reshape(5, 1) + big
    for x, y in ((m, k), (m, m.T.copy()), (big, oth), (m, np.arange(-2, 3, dtype=np.int64).reshape(5, 1)), (m.astype(np.float32), big)):
        with np.errstate(all="ignore"):
            ref = np.matmul(x, y)
        got = nd.matmul(nd.asarray(x), nd.asarray(y))
        assert got.dtype == ref.dtype, (x.dtype, y.dtype, got.dtype, ref.dtype)
        assert np.array_equal(got.get(), ref) if ref.dtype.kind in "iu" else np.allclose(got.get(), ref, rtol=1e-6), (x.dtype, y.dtype)
    assert np.array_equal(nd.tensordot(nd.asarray(m), nd.asarray(k.T.copy()), axes=1).get(), np.tensordot(m, k.T.copy(), axes=1))
    z = big.copy(); z[2] = 0
    assert all(np.array_equal(g.get(), r) for g, r in zip(nd.nonzero(nd.asarray(z)), np.nonzero(z)))
    assert np.array_equal(nd.isin(d, o).get(), np.isin(big, oth)) and np.array_equal(nd.isin(d, o, invert=True).get(), np.isin(big, oth, invert=True))
    assert np.array_equal(nd.mean(d, dtype=np.int64).get(), np.mean(big, dtype=np.int64))    # an integer dtype=: NumPy casts (wraps) first


NATIVE_CASES = [("multiply", np.int8, np.int8), ("add", np.uint8, np.uint8), ("subtract", np.int16, np.int16), ("less", np.uint16, np.uint16),
                ("maximum", np.uint32, np.uint32), ("add", np.float16, np.float16), ("floor_divide", np.uint64, np.uint64),
                ("add", np.int8, np.int32), ("true_divide", np.int8, np.int8), ("multiply", np.float16, np.float32)]


@pytest.mark.parametrize("name,ta,tb", NATIVE_CASES, ids=lambda v: getattr(v, "__name__", str(v)))
def test_native_narrow_calls_are_one_launch(lib, name, ta, tb, monkeypatch):
    """arithmetic on a storage-only dtype is ONE C-ABI call (mdhip_binary / mdhip_unary / mdhip_reduce with the
    narrow dtype codes) — no mdhip_convert before or after it, no wide temporaries."""
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(5)
    a, b = rng.integers(1, 100, 1000).astype(ta), rng.integers(1, 100, 1000).astype(tb)
    da, db = nd.asarray(a), nd.asarray(b)
    calls = {}

    def counting(entry):
        plain = getattr(lib, entry)

        def f(*args):
            calls[entry] = calls.get(entry, 0) + 1
            return plain(*args)
        return f

    for entry in ("binary", "unary", "convert", "reduce", "where"):
        monkeypatch.setattr(lib, entry, counting(entry))
    with np.errstate(all="ignore"):
        ref = getattr(np, name)(a, b)
    got = getattr(nd, name)(da, db)
    assert calls == {"binary": 1}, calls
    assert got.dtype == ref.dtype
    g = got.get()
    assert np.array_equal(g, ref) if ref.dtype.kind != "f" else np.allclose(g, ref, rtol=2e-3)
    calls.clear()
    s = nd.sum(da)
    m = nd.negative(da) if np.dtype(ta).kind != "u" else nd.absolute(da)
    assert calls == {"reduce": 1, "unary": 1}, calls
    with np.errstate(all="ignore"):
        assert s.dtype == a.sum().dtype and np.allclose(float(s.get()), float(a.sum()), rtol=2e-3)
        assert np.array_equal(m.get(), -a if np.dtype(ta).kind != "u" else np.abs(a))


def _strided_and_mixed(nd):
    """The generic narrow kernel: strided / broadcast operands and mixed storage types in one launch, bit-exact for integers."""
    rng = np.random.default_rng(9)
    for ta, tb in ((np.int8, np.uint8), (np.int16, np.int8), (np.uint16, np.uint32), (np.uint8, np.float16), (np.int8, np.int64), (np.float16, np.float64),
                   (np.uint64, np.uint8), (np.bool_, np.int8)):
        a = rng.integers(0, 2 if ta is np.bool_ else 120, (6, 8)).astype(ta)
        b = rng.integers(1, 120, (8, 6)).astype(tb)
        da, db = nd.asarray(a), nd.asarray(b)
        for name in ("add", "subtract", "multiply", "floor_divide", "maximum", "less_equal", "not_equal"):
            if ta is np.bool_ and name == "subtract":
                continue
            with np.errstate(all="ignore"):
                ref = getattr(np, name)(a[::2, 1:], b.T[::2, 1:])
                ref_b = getattr(np, name)(a, b[:, 0])
            got = getattr(nd, name)(da[::2, 1:], db.T[::2, 1:])
            got_b = getattr(nd, name)(da, db[:, 0])
            for g, r in ((got, ref), (got_b, ref_b)):
                assert g.dtype == r.dtype, (ta, tb, name, g.dtype, r.dtype)
                assert np.array_equal(g.get(), r) if r.dtype.kind != "f" else np.allclose(g.get(), r, rtol=2e-3), (ta, tb, name)
    # float16: a weak Python float is a float16 VALUE in the loop (0.1 -> 0.0999755859375)
    h = rng.standard_normal(64).astype(np.float16)
    dh = nd.asarray(h)
    for name, s in (("add", 0.1), ("multiply", 1.1), ("subtract", 3), ("true_divide", 0.3), ("power", 2.0)):
        ref, got = getattr(np, name)(h, s), getattr(nd, name)(dh, s)
        assert got.dtype == np.float16 and np.array_equal(got.get().view(np.uint16), ref.view(np.uint16)), (name, s)
    for name in ("sin", "exp", "sqrt", "tanh"):
        with np.errstate(all="ignore"):
            ref, got = getattr(np, name)(h), getattr(nd, name)(dh)
        assert got.dtype == np.float16 and np.allclose(got.get().astype(np.float32), ref.astype(np.float32), rtol=2e-3, atol=1e-3, equal_nan=True), name
    # in-place forms, incl. a loop whose result dtype differs from the destination's (same-kind cast back)
    u8 = rng.integers(0, 200, 50).astype(np.uint8)
    i16 = rng.integers(-300, 300, 50).astype(np.int16)
    c = nd.asarray(u8).copy(); c += nd.asarray(u8); r = u8.copy(); r += u8
    assert np.array_equal(c.get(), r)
    c = nd.asarray(i16).copy(); c *= nd.asarray(u8); r = i16.copy(); r *= u8
    assert np.array_equal(c.get(), r)
    # large contiguous arrays: the streaming kernels (16 B per lane), every element width, odd lengths (vector tail)
    for dt in NARROW:
        n = 70001
        a, b = rng.integers(1, 100, n).astype(dt), rng.integers(1, 100, n).astype(dt)
        da, db = nd.asarray(a), nd.asarray(b)
        for name in ("add", "multiply", "subtract", "greater", "mod", "minimum"):
            with np.errstate(all="ignore"):
                ref = getattr(np, name)(a, b)
            got = getattr(nd, name)(da, db).get()
            assert np.array_equal(got, ref) if ref.dtype.kind != "f" else np.allclose(got, ref, rtol=2e-3), (dt, name)
        with np.errstate(all="ignore"):
            assert np.array_equal(nd.multiply(da, 3).get(), a * 3) if np.dtype(dt).kind != "f" else np.allclose(nd.multiply(da, 3).get(), a * 3, rtol=2e-3)
            assert np.array_equal(nd.subtract(100, da).get(), 100 - a) if np.dtype(dt).kind != "f" else np.allclose(nd.subtract(100, da).get(), 100 - a, rtol=2e-3)
            assert np.array_equal(nd.negative(da).get(), -a) if np.dtype(dt).kind == "i" else True
            for red in ("sum", "max", "min", "argmax", "any"):
                ref, got = getattr(np, red)(a), getattr(nd, red)(da)
                assert got.dtype == ref.dtype and np.allclose(float(got.get()), float(ref), rtol=2e-3), (dt, red)


def test_narrow_strided_and_mixed_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _strided_and_mixed(nd)


@gpu
def test_narrow_strided_and_mixed_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _strided_and_mixed(nd)


def _index_and_compare(nd):
    """Gathers / scatters by index arrays in the storage-only types themselves (csrc/index.hip: no promote -> demote round trip),
    sizes on both sides of the serial / ordered split; 64-bit comparisons across signedness; Python ints beyond the loop dtype."""
    rng = np.random.default_rng(77)
    for dt in NARROW:
        kind = np.dtype(dt).kind
        # (three columns: element-granular order — serial up to 128 positions, sorted by destination above; sixteen columns: whole
        # rows, ordered at row granularity)
        for n, m, w in ((40, 25, 3), (3000, 9000, 3), (300, 6000, 16)):
            if kind == "f":
                a = (rng.standard_normal((n, w)) * 8).astype(dt)
                v = (rng.standard_normal((m, w)) * 8).astype(dt)
            else:
                info = np.iinfo(dt)
                a = rng.integers(info.min, info.max, (n, w), dtype=dt, endpoint=True)      # the whole range: uint64 >= 2**63 included
                v = rng.integers(info.min, info.max, (m, w), dtype=dt, endpoint=True)
            idx = rng.integers(-n, n, m)                                                    # duplicates, negative indices
            da, dv, di = nd.asarray(a), nd.asarray(v), nd.asarray(idx)
            assert np.array_equal(da[di].get(), a[idx]), (dt, n, "getitem")
            assert np.array_equal(da[di, 1].get(), a[idx, 1]), (dt, n, "getitem col")
            c, r = da.copy(), a.copy()
            c[di] = dv; r[idx] = v
            assert np.array_equal(c.get(), r), (dt, n, "setitem")                            # duplicates: the LAST write wins
            c, r = da.copy(), a.copy()
            with np.errstate(all="ignore"):
                nd.index_add(c, di, dv); np.add.at(r, idx, v)
            # (integers wrap; float16 rounds after EVERY contribution, in index order — as NumPy's unbuffered loop does)
            assert np.array_equal(c.get(), r, equal_nan=(kind == "f")), (dt, n, "index_add")
            c, r = da.copy(), a.copy()
            with np.errstate(all="ignore"):
                # (a typed scalar: for a bare Python int against uint64, NumPy's ufunc.at detours through float64 and drops low bits)
                nd.index_add(c, di, dt(3)); np.add.at(r, idx, dt(3))
            assert np.array_equal(c.get(), r, equal_nan=(kind == "f")), (dt, n, "index_add scalar")
            ai = rng.integers(0, n, (n, w))
            assert np.array_equal(nd.take_along_axis(da, nd.asarray(ai), 0).get(), np.take_along_axis(a, ai, 0)), (dt, n, "take_along_axis")
            c, r = da.copy(), a.copy()
            pi = np.argsort(rng.random((n, w)), axis=0)[: n // 2]
            nd.put_along_axis(c, nd.asarray(pi), dv[: n // 2], 0); np.put_along_axis(r, pi, v[: n // 2], 0)
            assert np.array_equal(c.get(), r), (dt, n, "put_along_axis")
        if kind in "iu":
            info = np.iinfo(dt)
            c = nd.asarray(np.zeros(4, dt))
            for bad in (info.max + 1, info.min - 1):
                with pytest.raises(OverflowError):
                    c[nd.asarray(np.array([0, 1]))] = bad                                  # NEP 50: a Python int must fit
            c[nd.asarray(np.array([0, 1]))] = info.max
            assert c.get()[1] == info.max
    # uint64 against signed operands: NumPy's 'Qq->?' loops compare the numbers, not the bits
    u = np.array([0, 5, 2 ** 63 + 5, 2 ** 64 - 1, 7], dtype=np.uint64)
    for sdt in (np.int64, np.int32, np.int8):
        sv = np.array([-1, 5, 100, -128, 8]).astype(sdt)
        for name in ("less", "less_equal", "greater", "greater_equal", "equal", "not_equal"):
            for x, y in ((u, sv), (sv, u), (u, sdt(-3)), (sdt(-3), u), (u[:1], sv), (u, sv[:1])):
                ref = getattr(np, name)(x, y)
                got = getattr(nd, name)(nd.asarray(x) if isinstance(x, np.ndarray) else x, nd.asarray(y) if isinstance(y, np.ndarray) else y)
                assert got.dtype == ref.dtype and np.array_equal(got.get(), ref), (sdt, name)
    # a Python int outside the loop dtype's range: comparisons answer for the number it is (arithmetic raises OverflowError)
    for arr in (u, np.array([1, -3, 7], dtype=np.int32), np.array([1, -3, 7], dtype=np.int8), np.array([1, -3], dtype=np.int64)):
        d = nd.asarray(arr)
        for py in (-1, -(2 ** 70), 2 ** 64 + 3, 2 ** 63, 2 ** 40, -129, 200, 5, 2 ** 63 + 5):
            for name in ("less", "less_equal", "greater", "greater_equal", "equal", "not_equal"):
                ref, ref_r = getattr(np, name)(arr, py), getattr(np, name)(py, arr)
                got, got_r = getattr(nd, name)(d, py), getattr(nd, name)(py, d)
                assert got.dtype == np.bool_ and np.array_equal(got.get(), ref), (arr.dtype, py, name)
                assert np.array_equal(got_r.get(), ref_r), (arr.dtype, py, name, "swapped")


def test_narrow_index_and_compare_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _index_and_compare(nd)


@gpu
def test_narrow_index_and_compare_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _index_and_compare(nd)


def test_half_conversion_rounds_to_nearest_even(lib):
    """md_double_to_half / md_half_to_double against NumPy over every binary16 value and the halfway cases around them."""
    from minidiff_amd import ndarray as nd
    allh = np.arange(0, 1 << 16, dtype=np.uint16).view(np.float16)
    finite = allh[np.isfinite(allh)]
    d = nd.asarray(finite)
    assert np.array_equal(d.astype(np.float64).get(), finite.astype(np.float64))            # widening is exact
    wide = finite.astype(np.float64)
    probes = np.concatenate([wide, wide * (1 + 2.0 ** -12), wide * (1 - 2.0 ** -12), np.nextafter(wide, np.inf), (wide[:-1] + wide[1:]) / 2,
                             [7e4, -7e4, 65519.9, 65520.0, 1e-8, 2.98e-8, 3e-8, np.inf, -np.inf]])
    with np.errstate(over="ignore"):
        ref = probes.astype(np.float16)
    got = nd.asarray(probes).astype(np.float16).get()
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    f32 = probes.astype(np.float32)
    with np.errstate(over="ignore"):
        ref32 = f32.astype(np.float16)
    assert np.array_equal(nd.asarray(f32).astype(np.float16).get().view(np.uint16), ref32.view(np.uint16))
    nan = nd.asarray(np.array([np.nan])).astype(np.float16).get()
    assert np.isnan(nan[0])
