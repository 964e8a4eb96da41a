"""Storage-only dtypes of the reference table (minidiff/backend/numpy.py:188-200: float16, int8/16, uint8/16/32/64) against
NumPy on seeded inputs: result dtypes, integer results bit-exact (wrap-around included), float16 within its precision; views,
transfers, conversions, gathers / scatters, in-place forms, creation functions. The reference-generated golden cases for
these dtypes are in tests/golden (narrow_*); minidiff_amd/narrow.py has the design."""
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


def test_uint64_upper_half_is_refused_loudly_where_values_matter(lib):
    from minidiff_amd import ndarray as nd
    big = np.array([1, 2 ** 63 + 5, 7], dtype=np.uint64)
    d = nd.asarray(big)
    assert np.array_equal(d.get(), big) and np.array_equal(d[::-1].get(), big[::-1])       # storage, views: fine
    assert np.array_equal(nd.concatenate([d, d]).get(), np.concatenate([big, big]))         # movers: fine (bits)
    assert np.array_equal(d.astype(np.float64).get(), big.astype(np.float64))              # conversion kernel: true values
    with pytest.raises(TypeError, match="uint64"):
        nd.max(d)
    with pytest.raises(TypeError, match="uint64"):
        nd.add(d, d)


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
