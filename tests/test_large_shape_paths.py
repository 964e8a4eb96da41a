"""Parity of the kernels and host-side re-expressions that serve LARGE, awkwardly laid-out
calls (scripts/shape_sweep.py found them walking memory badly): vectorised argmax/argmin over
rows and over columns (incl. ties and NaNs: first occurrence, as np.argmax), 16-byte run gathers,
the tiled transposing copy, reductions of permuted-dense views and of separated axis groups.
Twins: the CPU double checks the host logic, the GPU run checks the HIP kernels."""
import numpy as np
import pytest

from minidiff_amd import ndarray as nd

gpu = pytest.mark.gpu


def _twin(fn):
    def cpu(lib, on_gpu):
        if on_gpu:
            pytest.skip("other twin")
        fn()

    def dev(lib, on_gpu):
        assert on_gpu
        fn()
    return cpu, gpu(dev)


def _arg():
    rng = np.random.default_rng(0)
    for dt in (np.float32, np.float64, np.int64, np.int32):
        for shape in ((300, 4096), (2048, 1024), (3, 70000), (1500, 260)):
            if np.dtype(dt).kind == "f":
                h = rng.standard_normal(shape).astype(dt)
                h[rng.integers(0, shape[0], 40), rng.integers(0, shape[1], 40)] = np.nan      # first NaN wins
                h[:, 7] = np.inf if shape[1] > 7 else h[:, 7]
            else:
                h = rng.integers(-5, 5, shape).astype(dt)                                      # many ties: first wins
            d = nd.asarray(h)
            for axis in (0, 1, None):
                assert np.array_equal(np.asarray(nd.argmax(d, axis=axis)), np.argmax(h, axis=axis)), (dt, shape, axis)
                assert np.array_equal(np.asarray(nd.argmin(d, axis=axis)), np.argmin(h, axis=axis)), (dt, shape, axis)
        # unaligned row starts (odd row length) and a sliced view
        h = rng.standard_normal((257, 4099)).astype(dt) if np.dtype(dt).kind == "f" else rng.integers(-9, 9, (257, 4099)).astype(dt)
        d = nd.asarray(h)
        assert np.array_equal(np.asarray(nd.argmax(d, axis=1)), np.argmax(h, axis=1))
        assert np.array_equal(np.asarray(nd.argmax(d[:, 3:], axis=1)), np.argmax(h[:, 3:], axis=1))
        assert np.array_equal(np.asarray(nd.argmin(d[1:, :4096], axis=0)), np.argmin(h[1:, :4096], axis=0))
    allneg = np.full((4, 2048), -np.inf, dtype=np.float32)
    assert np.array_equal(np.asarray(nd.argmax(nd.asarray(allneg), axis=1)), np.zeros(4, dtype=np.int64))


test_arg_cpu, test_arg_gpu = _twin(_arg)


def _gather():
    rng = np.random.default_rng(1)
    for dt in (np.float32, np.float64, np.int64, np.bool_):
        h = (rng.standard_normal((500, 256)) * 10).astype(dt)
        d = nd.asarray(h)
        idx = rng.integers(-500, 500, (333,))
        assert np.array_equal(np.asarray(d[nd.asarray(idx)]), h[idx])                         # rows: 16-B runs
        idx2 = rng.integers(0, 500, (7, 11))
        assert np.array_equal(np.asarray(d[nd.asarray(idx2)]), h[idx2])
        assert np.array_equal(np.asarray(d[nd.asarray(idx), 4:]), h[idx, 4:])                 # still aligned runs (f32: 16 B)
        assert np.array_equal(np.asarray(d[nd.asarray(idx), 3:]), h[idx, 3:])                 # unaligned: element path
        assert np.array_equal(np.asarray(d[:, nd.asarray(idx[:100] % 256)]), h[:, idx[:100] % 256])   # index on the last axis
        t = nd.asarray(np.ascontiguousarray(h.reshape(50, 10, 256)))
        i3 = rng.integers(0, 10, (6,))
        assert np.array_equal(np.asarray(t[:, nd.asarray(i3)]), h.reshape(50, 10, 256)[:, i3])
    d = nd.asarray(np.zeros((500, 256), dtype=np.float32))
    with pytest.raises(IndexError):
        d[nd.asarray(np.array([0, 500] * 40))]


test_gather_cpu, test_gather_gpu = _twin(_gather)


def _transposed():
    rng = np.random.default_rng(2)
    for dt in (np.float32, np.float64, np.int64):
        h = (rng.standard_normal((300, 1000)) * 7).astype(dt)
        w = (rng.standard_normal((1000, 300)) * 7).astype(dt)
        d, e = nd.asarray(h), nd.asarray(w)
        assert np.array_equal(np.asarray(nd.copy(d.T)), h.T)
        assert np.array_equal(np.asarray(nd.add(e, d.T)), w + h.T)                            # straightened operand
        assert np.array_equal(np.asarray(nd.multiply(d.T, d.T)), h.T * h.T)
        assert np.array_equal(np.asarray(nd.negative(d.T)), -h.T)
        t = (rng.standard_normal((5, 130, 70)) * 3).astype(dt)
        td = nd.asarray(t)
        assert np.array_equal(np.asarray(nd.copy(nd.swapaxes(td, 1, 2))), np.swapaxes(t, 1, 2))   # batched
        assert np.array_equal(np.asarray(nd.copy(td.T)), t.T)                                      # full reversal: generic
        assert np.array_equal(np.asarray(nd.copy(d[:, ::3].T)), h[:, ::3].T)                       # strided both ways
        if np.dtype(dt).kind == "f":
            np.testing.assert_allclose(np.asarray(nd.exp(nd.multiply(d.T, 0.01))), np.exp(h.T * dt(0.01)), rtol=1e-6)
            assert np.asarray(nd.astype(d.T, np.float64)).dtype == np.float64
            assert np.array_equal(np.asarray(nd.astype(d.T, np.float64)), h.T.astype(np.float64))


test_transposed_cpu, test_transposed_gpu = _twin(_transposed)


def _staged():
    rng = np.random.default_rng(3)
    h = rng.standard_normal((40, 130, 96)).astype(np.float32)
    d = nd.asarray(h)
    hi = rng.integers(-50, 50, (40, 130, 96))
    di = nd.asarray(hi)
    for axis in ((0, 2), (2, 0), (0, 1), (1, 2), 1, None):
        for keep in (False, True):
            ref = h.astype(np.float64).sum(axis=axis, keepdims=keep)          # f32 accumulation: error scales with sum|x|
            bound = 2e-6 * np.abs(h).astype(np.float64).sum(axis=axis, keepdims=keep).max()
            assert np.abs(np.asarray(nd.sum(d, axis=axis, keepdims=keep)) - ref).max() <= bound
            assert np.array_equal(np.asarray(nd.max(d, axis=axis, keepdims=keep)), h.max(axis=axis, keepdims=keep))
            assert np.array_equal(np.asarray(nd.sum(di, axis=axis, keepdims=keep)), hi.sum(axis=axis, keepdims=keep))   # ints: exact
            assert np.array_equal(np.asarray(nd.any(nd.greater(d, 3.5), axis=axis, keepdims=keep)), (h > 3.5).any(axis=axis, keepdims=keep))
    np.testing.assert_allclose(np.asarray(nd.mean(d, axis=(0, 2))), h.astype(np.float64).mean(axis=(0, 2)), rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.asarray(nd.std(d, axis=(0, 2))), h.std(axis=(0, 2)), rtol=2e-5)
    q = rng.standard_normal((12, 20, 1, 33, 37)).astype(np.float64)      # extent-1 axis between two reduced ones
    qd = nd.asarray(q)
    np.testing.assert_allclose(np.asarray(nd.sum(qd, axis=(0, 1, 3))), q.sum(axis=(0, 1, 3)), rtol=1e-12)
    np.testing.assert_allclose(np.asarray(nd.sum(qd, axis=(0, 3, 4))), q.sum(axis=(0, 3, 4)), rtol=1e-12)
    # permuted-dense views reduced over everything
    for view, ref in ((d.T, h.T), (nd.swapaxes(d, 0, 1), np.swapaxes(h, 0, 1)), (nd.transpose(d, (1, 2, 0)), np.transpose(h, (1, 2, 0)))):
        np.testing.assert_allclose(float(np.asarray(nd.sum(view))), float(ref.astype(np.float64).sum()), rtol=1e-5)
        assert float(np.asarray(nd.max(view))) == ref.max()
        assert np.asarray(nd.sum(view, keepdims=True)).shape == (1, 1, 1)
    assert int(np.asarray(nd.sum(di.T))) == int(hi.sum())
    np.testing.assert_allclose(float(np.asarray(nd.sum(d[:, ::2].T))), float(h[:, ::2].astype(np.float64).sum()), rtol=1e-5)   # gaps: not dense
    assert nd.argmax(d.T).item() == int(np.argmax(h.T))     # order matters: never flattened


test_staged_cpu, test_staged_gpu = _twin(_staged)


def _casts():
    rng = np.random.default_rng(4)
    dts = (np.float32, np.float64, np.int64, np.int32, np.bool_)
    for shape in ((300, 1000), (70001,), (5, 7, 44)):
        for src in dts:
            h = (rng.standard_normal(shape) * 50).astype(src)
            d = nd.asarray(h)
            for dst in dts:
                got = np.asarray(nd.astype(d, dst))
                assert got.dtype == np.dtype(dst)
                assert np.array_equal(got, h.astype(dst)), (shape, src, dst)
            assert np.array_equal(np.asarray(nd.astype(d[1:], np.float64)), h[1:].astype(np.float64))   # unaligned start


test_casts_cpu, test_casts_gpu = _twin(_casts)


def _scatter():
    """np.add.at / a[idx] = v at row granularity: duplicates accumulate in index order (bit-exact
    float sums), the last duplicate wins for assignment."""
    rng = np.random.default_rng(6)
    for dt in (np.float32, np.float64, np.int64):
        for n_idx, hi in ((3000, 40), (2000, 5000), (1500, 1)):       # heavy duplication / nearly unique / all equal
            R, Cn = 5000 if hi > 100 else 64, 48
            base = (rng.standard_normal((R, Cn)) * 3).astype(dt)
            vals = (rng.standard_normal((n_idx, Cn)) * 3).astype(dt)
            idx = rng.integers(-hi, hi, (n_idx,)) if hi <= R else rng.integers(0, R, (n_idx,))
            exp = base.copy()
            np.add.at(exp, idx, vals)
            d = nd.asarray(base)
            nd.index_add(d, nd.asarray(idx), nd.asarray(vals))
            assert np.array_equal(np.asarray(d), exp), (dt, n_idx, hi)
            exp = base.copy()
            exp[idx] = vals
            d = nd.asarray(base)
            d[nd.asarray(idx)] = nd.asarray(vals)
            assert np.array_equal(np.asarray(d), exp), (dt, n_idx, hi, "set")
        base = (rng.standard_normal((300, 64)) * 3).astype(dt)
        idx = rng.integers(0, 300, (500,))
        for vals in ((rng.standard_normal((500, 1)) * 3).astype(dt), (rng.standard_normal((1, 64)) * 3).astype(dt), dt(2.5) if dt != np.int64 else 3):
            exp = base.copy()
            np.add.at(exp, idx, vals)
            d = nd.asarray(base)
            nd.index_add(d, nd.asarray(idx), nd.asarray(vals) if isinstance(vals, np.ndarray) else vals)
            assert np.array_equal(np.asarray(d), exp)
        # a slice of the last axis (run shorter than the row), and an index on a middle axis
        exp = base.copy()
        vals = (rng.standard_normal((500, 37)) * 3).astype(dt)
        np.add.at(exp[:, 5:42], idx, vals)
        d = nd.asarray(base)
        nd.index_add(d[:, 5:42], nd.asarray(idx), nd.asarray(vals))
        assert np.array_equal(np.asarray(d), exp)
        t = (rng.standard_normal((20, 30, 40)) * 3).astype(dt)
        i3 = rng.integers(0, 30, (200,))
        v3 = (rng.standard_normal((20, 200, 40)) * 3).astype(dt)
        exp = t.copy()
        np.add.at(exp, (slice(None), i3), v3)
        d = nd.asarray(t)
        nd.index_add(d, (slice(None), nd.asarray(i3)), nd.asarray(v3))
        assert np.array_equal(np.asarray(d), exp)
    # many tiles of the stable radix sort behind the row-granular path (index.hip: 2048 rows per tile, 8 bits per pass):
    # 300k contributions, offsets up to 800k elements (3 passes), heavy and light duplication, and a destination
    # walked backwards (negative row stride: keys are offsets relative to the smallest reachable one)
    for R, hi in ((100_000, 100_000), (100_000, 37), (257, 257)):
        base = (rng.standard_normal((R, 8)) * 3).astype(np.float32)
        idx = rng.integers(0, hi, (300_000,))
        vals = (rng.standard_normal((300_000, 8)) * 3).astype(np.float32)
        exp = base.copy()
        np.add.at(exp, idx, vals)
        d = nd.asarray(base)
        nd.index_add(d, nd.asarray(idx), nd.asarray(vals))
        assert np.array_equal(np.asarray(d), exp), (R, hi)
        exp = base.copy()
        np.add.at(exp[::-1], idx, vals)
        d = nd.asarray(base)
        nd.index_add(d[::-1], nd.asarray(idx), nd.asarray(vals))
        assert np.array_equal(np.asarray(d), exp), (R, hi, "reversed")
        exp = base.copy()
        exp[idx] = vals
        d = nd.asarray(base)
        d[nd.asarray(idx)] = nd.asarray(vals)
        assert np.array_equal(np.asarray(d), exp), (R, hi, "set")
    d = nd.asarray(np.zeros((50, 64), dtype=np.float32))
    with pytest.raises(IndexError):
        nd.index_add(d, nd.asarray(np.array([0, 50] * 100)), nd.asarray(np.ones((200, 64), dtype=np.float32)))


test_scatter_cpu, test_scatter_gpu = _twin(_scatter)


def _ragged_matmul(shapes=((1000, 784, 512), (257, 130, 515), (1031, 1028, 260), (300, 17, 301), (2000, 1000, 1500))):
    """Shapes that are not multiples of the block tile (guarded edge variant of the MFMA kernel:
    whole 16-B loads wherever a vector lies inside the operand), all three layouts of the
    matmul backward, aligned and unaligned row strides."""
    rng = np.random.default_rng(7)
    for (M, K, N) in shapes:
        a = rng.standard_normal((M, K)).astype(np.float32)
        b = rng.standard_normal((K, N)).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        scale = np.abs(ref).max()
        A, B = nd.asarray(a), nd.asarray(b)
        At, Bt = nd.asarray(np.ascontiguousarray(a.T)), nd.asarray(np.ascontiguousarray(b.T))
        for x, y in ((A, B), (A, Bt.T), (At.T, B), (At.T, Bt.T)):
            got = np.asarray(nd.matmul(x, y))
            assert got.shape == (M, N)
            assert np.abs(got - ref).max() / scale < 2e-6, (M, K, N)
        # views with an offset (rows no longer 16-B aligned) and a batch
        got = np.asarray(nd.matmul(A[1:, 1:], B[1:, 3:]))
        assert np.abs(got - a[1:, 1:].astype(np.float64) @ b[1:, 3:].astype(np.float64)).max() / scale < 2e-6
    t = rng.standard_normal((3, 200, 96)).astype(np.float32)
    u = rng.standard_normal((3, 96, 130)).astype(np.float32)
    got = np.asarray(nd.matmul(nd.asarray(t), nd.asarray(u)))
    assert np.abs(got - t.astype(np.float64) @ u.astype(np.float64)).max() / np.abs(got).max() < 2e-6


_, test_ragged_matmul_gpu = _twin(_ragged_matmul)


@gpu
def test_skinny_matmul_split_k_gpu(lib, on_gpu):
    """Few output tiles and a long k (small batch through a wide layer): the grid is widened along
    k, partial products are summed in split order — same result on every run, within fp32 of f64."""
    assert on_gpu
    rng = np.random.default_rng(8)
    for (M, K, N) in ((64, 4096, 4096), (4096, 4096, 64), (32, 2048, 300), (100, 5000, 70), (128, 1024, 128)):
        a = rng.standard_normal((M, K)).astype(np.float32)
        b = rng.standard_normal((K, N)).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        A, B = nd.asarray(a), nd.asarray(b)
        At, Bt = nd.asarray(np.ascontiguousarray(a.T)), nd.asarray(np.ascontiguousarray(b.T))
        first = None
        for x, y in ((A, B), (A, Bt.T), (At.T, B)):
            got = np.asarray(nd.matmul(x, y))
            assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6, (M, K, N)
            if first is None:
                first = got
                assert np.array_equal(np.asarray(nd.matmul(x, y)), first)      # deterministic
    t = rng.standard_normal((2, 64, 2048)).astype(np.float32)
    u = rng.standard_normal((2, 2048, 128)).astype(np.float32)
    got = np.asarray(nd.matmul(nd.asarray(t), nd.asarray(u)))
    ref = t.astype(np.float64) @ u.astype(np.float64)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6


@gpu
def test_thin_products_streaming_kernels_gpu(lib, on_gpu):
    """Matrix x vector, vector x matrix and up to eight columns / rows (skinny.hip: one read of the big operand, no matrix cores),
    both storage orders of the big operand, 1-D and 2-D thin operands, batches, float32 and float64, sizes the 16-B lanes do not
    divide evenly into strips. Integer-valued operands: the products must EQUAL NumPy's; repeated calls bit-identical."""
    assert on_gpu
    rng = np.random.default_rng(88)
    for dtype in (np.float32, np.float64):
        for (R, K) in ((1024, 1024), (2052, 516), (4096, 260), (8192, 128), (520, 2048)):
            a = rng.integers(-3, 4, (R, K)).astype(dtype)
            A, At = nd.asarray(a), nd.asarray(np.ascontiguousarray(a.T))
            for nc in (1, 2, 3, 5, 8):
                v = rng.integers(-3, 4, (K, nc)).astype(dtype)
                u = rng.integers(-3, 4, (nc, R)).astype(dtype)
                V, Vt, U = nd.asarray(v), nd.asarray(np.ascontiguousarray(v.T)), nd.asarray(u)
                ref = a.astype(np.float64) @ v
                for x in (A, At.T):                       # big operand row-major / transposed view
                    for y in (V, Vt.T):                   # thin operand as columns / as k-contiguous rows
                        got = nd.matmul(x, y)
                        assert got.dtype == dtype and np.array_equal(got.get(), ref), (dtype, R, K, nc)
                ref2 = u.astype(np.float64) @ a
                for x in (A, At.T):
                    got = nd.matmul(U, x)
                    assert np.array_equal(got.get(), ref2), (dtype, R, K, nc, "thin M")
                    assert np.array_equal(nd.matmul(U, x).get(), got.get())
            v1 = rng.integers(-3, 4, (K,)).astype(dtype)
            assert np.array_equal(nd.matmul(A, nd.asarray(v1)).get(), a.astype(np.float64) @ v1)
            w1 = rng.integers(-3, 4, (R,)).astype(dtype)
            assert np.array_equal(nd.matmul(nd.asarray(w1), A).get(), w1.astype(np.float64) @ a)
    # batched, and a strided thin operand (a column of a wider matrix)
    t = rng.integers(-3, 4, (3, 1024, 512)).astype(np.float32)
    wide = rng.integers(-3, 4, (3, 512, 10)).astype(np.float32)
    got = nd.matmul(nd.asarray(t), nd.asarray(wide)[:, :, 3:5])
    assert np.array_equal(got.get(), t.astype(np.float64) @ wide[:, :, 3:5])
    got = nd.matmul(nd.swapaxes(nd.asarray(wide)[:, :, 3:5], -1, -2), nd.swapaxes(nd.asarray(t), -1, -2))
    assert np.array_equal(got.get(), np.swapaxes(wide[:, :, 3:5], -1, -2).astype(np.float64) @ np.swapaxes(t, -1, -2))
    # float data at full size against float64
    a = rng.standard_normal((8192, 4096)).astype(np.float32)
    v = rng.standard_normal((4096, 1)).astype(np.float32)
    ref = a.astype(np.float64) @ v
    for x in (nd.asarray(a), nd.asarray(np.ascontiguousarray(a.T)).T):
        got = nd.matmul(x, nd.asarray(v)).get()
        assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6

@gpu
def test_any_all_of_whole_arrays_gpu(lib, on_gpu):
    """any / all over a whole contiguous array (reduce.hip, k_anyall_flat): truth values from 16-B vectors of the array's own type.
    One true among zeros / one zero among ones at the first, a middle, the last vector and in the scalar tail; NaN is true, -0.0 is
    false; every dtype; sizes that are not whole vectors; several axes of an N-d array."""
    assert on_gpu
    rng = np.random.default_rng(99)
    for dtype in (np.bool_, np.int32, np.int64, np.float32, np.float64):
        for n in (1 << 16, (1 << 20) + 7, 3_000_001):
            for pos in (0, n // 2 + 3, n - 17, n - 1):
                h = np.zeros(n, dtype=dtype)
                assert not bool(nd.any(nd.asarray(h)).item()) and not bool(nd.all(nd.asarray(h)).item())
                h[pos] = 1
                assert bool(nd.any(nd.asarray(h)).item()) and not bool(nd.all(nd.asarray(h)).item()), (dtype, n, pos)
                h = np.ones(n, dtype=dtype)
                assert bool(nd.all(nd.asarray(h)).item()) and bool(nd.any(nd.asarray(h)).item())
                h[pos] = 0
                assert not bool(nd.all(nd.asarray(h)).item()) and bool(nd.any(nd.asarray(h)).item()), (dtype, n, pos)
    f = np.zeros(1 << 17, dtype=np.float32)
    f[12345] = -0.0
    assert not bool(nd.any(nd.asarray(f)).item())
    f[54321] = np.nan
    assert bool(nd.any(nd.asarray(f)).item())
    g = np.full(1 << 17, np.nan, dtype=np.float64)
    assert bool(nd.all(nd.asarray(g)).item())
    t = rng.standard_normal((64, 33, 129)).astype(np.float32)
    d = nd.asarray(t)
    assert bool(nd.any(nd.isnan(d)).item()) is False and bool(nd.all(nd.less(d, 100.0)).item()) is True
    t[63, 32, 128] = np.nan
    assert bool(nd.any(nd.isnan(nd.asarray(t))).item()) is True
    assert np.array_equal(nd.any(nd.asarray(t > 3), axis=(0, 1, 2), keepdims=True).get(), np.any(t > 3, axis=(0, 1, 2), keepdims=True))


def test_ragged_matmul_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    _ragged_matmul(shapes=((257, 130, 515), (300, 17, 301)))   # the CPU double's GEMM is a plain triple loop


@gpu
def test_f64_matmul_mfma_gpu(lib, on_gpu):
    """float64 matmul on the f64 matrix cores: the three layouts of the matmul backward, aligned,
    ragged and offset operands, batched; exact integer data pins the lane maps."""
    assert on_gpu
    rng = np.random.default_rng(9)
    for (M, K, N) in ((256, 256, 256), (1024, 512, 768), (130, 70, 515), (64, 64, 64), (2048, 64, 2048), (10, 30, 20), (300, 17, 301)):
        a = rng.standard_normal((M, K))
        b = rng.standard_normal((K, N))
        ref = a @ b
        A, B = nd.asarray(a), nd.asarray(b)
        At, Bt = nd.asarray(np.ascontiguousarray(a.T)), nd.asarray(np.ascontiguousarray(b.T))
        for x, y in ((A, B), (A, Bt.T), (At.T, B), (At.T, Bt.T)):
            got = np.asarray(nd.matmul(x, y))
            assert got.dtype == np.float64 and got.shape == (M, N)
            assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-14, (M, K, N)
        if M > 64:
            got = np.asarray(nd.matmul(A[1:, 1:], B[1:, 3:]))
            assert np.abs(got - a[1:, 1:] @ b[1:, 3:]).max() / np.abs(ref).max() < 1e-14
        ai = rng.integers(-8, 8, (M, K)).astype(np.float64)        # exact in any order: every element must match
        bi = rng.integers(-8, 8, (K, N)).astype(np.float64)
        assert np.array_equal(np.asarray(nd.matmul(nd.asarray(ai), nd.asarray(bi))), ai @ bi)
    t = rng.standard_normal((3, 200, 96))
    u = rng.standard_normal((3, 96, 130))
    got = np.asarray(nd.matmul(nd.asarray(t), nd.asarray(u)))
    assert np.abs(got - t @ u).max() / np.abs(got).max() < 1e-14


@gpu
def test_f64_tn_direct_to_lds_gpu(lib, on_gpu, mdopt):
    """k_gemm_f64_tn_glds (TN products of whole aligned 128 x 128 x 16 tiles on grids of >= 256 tiles): integer-valued operands,
    so every element must EQUAL NumPy's — whole arrays, batched, a sub-view with a larger row stride — with the direct-to-LDS
    kernel on and off (option gemm_glds, mdhip_debug_set_option), and the random-data result within 1e-14 of NumPy's."""
    assert on_gpu
    rng = np.random.default_rng(19)
    try:
        for (M, K, N, batch) in ((2048, 96, 2048, 0), (2048, 512, 2048, 0), (1024, 64, 1024, 4), (4096, 32, 2048, 0)):
            lead = (batch,) if batch else ()
            big_a = rng.integers(-4, 5, lead + (K, M + 8)).astype(np.float64)      # (row stride M + 8: an aligned sub-view)
            big_b = rng.integers(-4, 5, lead + (K, N)).astype(np.float64)
            da, db = nd.asarray(big_a)[..., :, 2:M + 2], nd.asarray(big_b)
            ha = big_a[..., :, 2:M + 2]
            ref = np.matmul(np.swapaxes(ha, -1, -2), big_b)
            for flag in ("1", "0"):
                mdopt("gemm_glds", flag)
                got = nd.matmul(nd.swapaxes(da, -1, -2), db).get()
                assert np.array_equal(got, ref), (M, K, N, batch, flag)
        mdopt("gemm_glds", 1)
        a, b = rng.standard_normal((512, 2048)), rng.standard_normal((512, 2048))
        got = nd.matmul(nd.asarray(a).T, nd.asarray(b)).get()
        ref = a.T @ b
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-14
    finally:
        mdopt("gemm_glds", 1)


def _heavy_duplicates(nd, n):
    """np.add.at / a[idx] = v with MANY positions per destination at element granularity (a histogram): positions sorted by
    destination, one serial pass per destination (csrc/index.hip scatter_sorted) — NumPy's accumulation order bit for bit, and no
    round trip per multiplicity level (a million contributions to one element used to be a million host-driven rounds)."""
    import time
    rng = np.random.default_rng(8)
    for dt in (np.float32, np.float64, np.float16, np.int8, np.int16, np.uint8, np.bool_, np.int32):
        kind = np.dtype(dt).kind
        bins = rng.integers(0, 10, n)                               # ten destinations
        vals = (rng.standard_normal(n) * 3).astype(dt) if kind == "f" else (rng.integers(0, 2, n).astype(dt) if kind == "b" else rng.integers(-5, 6, n).astype(dt))
        a = np.zeros(16, dt)
        d = nd.asarray(a)
        t0 = time.perf_counter()
        nd.index_add(d, nd.asarray(bins), nd.asarray(vals))
        got = d.get()
        dt_s = time.perf_counter() - t0
        with np.errstate(all="ignore"):
            np.add.at(a, bins, vals)
        assert np.array_equal(got, a, equal_nan=(kind == "f")), (dt, got, a)
        assert dt_s < 20.0, (dt, dt_s)
        # last write wins, in plan order
        a2 = np.zeros(16, dt); d2 = nd.asarray(a2)
        d2[nd.asarray(bins)] = nd.asarray(vals); a2[bins] = vals
        assert np.array_equal(d2.get(), a2, equal_nan=(kind == "f")), dt
    # two index arrays, strided destination, scalar value; every position on ONE element
    a = np.zeros((7, 9), np.float32); d = nd.asarray(a)
    i0, i1 = rng.integers(0, 7, n), rng.integers(0, 9, n)
    nd.index_add(d[::-1, ::2], (nd.asarray(i0 % 7), nd.asarray(i1 % 5)), np.float32(0.1))
    np.add.at(a[::-1, ::2], (i0 % 7, i1 % 5), np.float32(0.1))
    assert np.array_equal(d.get(), a)
    one = np.zeros(3, np.float32); done = nd.asarray(one)
    v = rng.standard_normal(n).astype(np.float32)
    nd.index_add(done, nd.asarray(np.full(n, 1)), nd.asarray(v))
    np.add.at(one, np.full(n, 1), v)
    assert np.array_equal(done.get(), one)


def test_heavy_duplicate_scatters_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _heavy_duplicates(nd, 20000)


@pytest.mark.gpu
def test_heavy_duplicate_scatters_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _heavy_duplicates(nd, 300000)


def _thin_by_thin_long_k(nd, K):
    """A few rows times a few columns over a LONG k (np.dot of two vectors, (1,K)@(K,1), (8,K)@(K,8), batched): k is cut over the
    chip and the block partials are added in block order (csrc/skinny.hip md_gemm_longk) — integers exact, floats within the
    summation-order bound, bit-identical run to run; transposed / strided operands by strides."""
    rng = np.random.default_rng(12)
    for dt in (np.float32, np.float64, np.int32, np.int64):
        def mk(*shape):
            return rng.standard_normal(shape).astype(dt) if np.dtype(dt).kind == "f" else rng.integers(-9, 10, shape).astype(dt)
        for M, N in ((1, 1), (2, 1), (1, 3), (4, 4), (8, 1), (1, 8), (8, 8), (3, 5), (16, 16), (9, 3), (1, 64), (64, 1), (12, 1), (1, 100), (64, 64), (128, 33)):
            if max(M, N) > 8 and ((K < 64 * max(M, N) and K > 10000) or K > 400000):
                continue        # (the tile kernels' territory / host-side reference too slow to be worth it)
            A, B = mk(M, K), mk(K, N)
            cases = [(nd.asarray(A), nd.asarray(B), A, B)]
            At, Bt = np.ascontiguousarray(A.T), np.ascontiguousarray(B.T)
            cases.append((nd.asarray(At).T, nd.asarray(Bt).T, At.T, Bt.T))                       # the other unit-stride axis
            for da, db, ha, hb in cases:
                got = nd.matmul(da, db)
                # (NumPy's integer matmul has no BLAS behind it — 17 s for 64 x 64 x 300,007; the float64 product of these small
                # integers is exact: |sum| <= 81 K < 2**53)
                ref = np.matmul(ha, hb) if np.dtype(dt).kind == "f" else np.matmul(ha.astype(np.float64), hb.astype(np.float64)).astype(dt)
                assert got.dtype == ref.dtype and got.shape == ref.shape
                if np.dtype(dt).kind == "f":
                    scale = np.abs(ha).astype(np.float64) @ np.abs(hb).astype(np.float64)
                    assert (np.abs(got.get().astype(np.float64) - ref) <= (2e-6 if dt is np.float32 else 1e-13) * scale + 1e-30).all(), (dt, M, N)
                    assert np.array_equal(nd.matmul(da, db).get(), got.get())                      # fixed summation order
                else:
                    assert np.array_equal(got.get(), ref), (dt, M, N)
        x, y = mk(K), mk(K)
        ref = np.dot(x, y)
        got = nd.dot(nd.asarray(x), nd.asarray(y))
        assert got.shape == () and got.dtype == ref.dtype
        assert np.array_equal(got.get(), ref) if np.dtype(dt).kind != "f" else abs(float(got.get()) - float(ref)) <= (2e-6 if dt is np.float32 else 1e-13) * float(np.abs(x).astype(np.float64) @ np.abs(y))
        xs, ys = mk(2 * K)[::2], mk(3 * K)[::3]                                                    # strided vectors
        ref = np.dot(xs, ys)
        got = nd.dot(nd.asarray(np.ascontiguousarray(np.repeat(xs, 2)))[::2], nd.asarray(ys.copy()))
        assert np.array_equal(got.get(), ref) if np.dtype(dt).kind != "f" else np.isclose(float(got.get()), float(ref), rtol=1e-3, atol=1e-2)
        Ab, Bb = mk(5, 2, K), mk(5, K, 3)                                                          # batched
        got, ref = nd.matmul(nd.asarray(Ab), nd.asarray(Bb)), (np.matmul(Ab, Bb) if np.dtype(dt).kind == "f" else np.matmul(Ab.astype(np.float64), Bb.astype(np.float64)).astype(dt))
        assert np.array_equal(got.get(), ref) if np.dtype(dt).kind != "f" else np.allclose(got.get(), ref, rtol=1e-4, atol=1e-2 if dt is np.float32 else 1e-8)


def test_thin_by_thin_long_k_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _thin_by_thin_long_k(nd, 9001)


@pytest.mark.gpu
def test_thin_by_thin_long_k_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    for K in (8192, 9001, 300007, 2_500_000):
        _thin_by_thin_long_k(nd, K)


def _long_strided_arg_lines(nd):
    """argmax / argmin down a handful of long STRIDED lines (the two columns of a 2,000,000 x 2 array): gathered into rows first
    (ndarray._arg_reduce) — same indices, first occurrence among ties, keepdims, views."""
    rng = np.random.default_rng(4)
    for shp, ax in (((70000, 2), 0), ((70000, 3), 0), ((2, 70000, 3), 1), ((70000, 2, 2), 0), ((3, 70000), 1), ((70000, 2), -2), ((200000, 5, 1), 0)):
        a = rng.integers(0, 50, shp).astype(np.float32)      # many ties
        for kd in (False, True):
            for f in ("argmax", "argmin"):
                r, e = getattr(nd, f)(nd.asarray(a), axis=ax, keepdims=kd).get(), getattr(np, f)(a, axis=ax, keepdims=kd)
                assert r.shape == e.shape and np.array_equal(r, e), (shp, ax, kd, f)
        assert np.array_equal(nd.argmax(nd.asarray(a)[::2], axis=ax).get(), np.argmax(a[::2], axis=ax))


def test_long_strided_arg_lines_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _long_strided_arg_lines(nd)


@pytest.mark.gpu
def test_long_strided_arg_lines_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _long_strided_arg_lines(nd)


def _mid_length_arg_rows(nd):
    """argmax / argmin over the last axis for rows of 24 .. 1023 elements (a wave per row, csrc/reduce.hip k_arg_rows_wave): first of
    equal values, the first NaN, every compute dtype, strided row starts, 3-D."""
    rng = np.random.default_rng(9)
    for rows, cols in ((300, 24), (1000, 64), (4097, 268), (700, 1023), (65, 100)):
        for dt in (np.float32, np.float64, np.int32, np.int64, np.bool_, np.int8, np.uint16, np.float16, np.uint64):
            a = (rng.random((rows, cols)) > 0.5) if dt is np.bool_ else rng.integers(-4 if np.dtype(dt).kind != "u" else 0, 5, (rows, cols)).astype(dt)      # many ties
            if np.dtype(dt).kind == "f":
                a[rng.integers(0, rows, rows // 3), rng.integers(0, cols, rows // 3)] = np.nan
            d = nd.asarray(a)
            for f in ("argmax", "argmin"):
                assert np.array_equal(getattr(nd, f)(d, axis=-1).get(), getattr(np, f)(a, axis=-1)), (rows, cols, dt, f)
                assert np.array_equal(getattr(nd, f)(d[::3], axis=1, keepdims=True).get(), getattr(np, f)(a[::3], axis=1, keepdims=True)), (rows, cols, dt, f)
    b = rng.integers(0, 3, (40, 50, 130)).astype(np.float32)
    assert np.array_equal(nd.argmax(nd.asarray(b), axis=2).get(), np.argmax(b, axis=2))
    assert np.array_equal(nd.argmin(nd.asarray(b)[:, ::2], axis=-1).get(), np.argmin(b[:, ::2], axis=-1))


def test_mid_length_arg_rows_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _mid_length_arg_rows(nd)


@pytest.mark.gpu
def test_mid_length_arg_rows_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _mid_length_arg_rows(nd)


@pytest.mark.gpu
def test_few_tiles_long_k_products_gpu(lib, on_gpu):
    """A few dozen output tiles under a long k (the weight gradient of a 1000-wide layer over a large batch): k ranges run as the batch
    of one launch, partials added in range order (csrc/gemm.hip HipExec::gemm). All three layouts, a k that does not divide (the
    remainder launch), integer-valued operands exact, a strided destination."""
    assert on_gpu
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(6)
    for M, K, N in ((1000, 40000, 1000), (1024, 32768 + 96, 768), (900, 20000, 1100)):
        A = rng.integers(-3, 4, (M, K)).astype(np.float32)
        B = rng.integers(-3, 4, (K, N)).astype(np.float32)
        ref = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)          # exact: |sum| < 2**24
        dA, dB = nd.asarray(A), nd.asarray(B)
        dAt, dBt = nd.asarray(np.ascontiguousarray(A.T)), nd.asarray(np.ascontiguousarray(B.T))
        for name, got in (("NN", nd.matmul(dA, dB)), ("NT", nd.matmul(dA, dBt.T)), ("TN", nd.matmul(dAt.T, dB)), ("TT", nd.matmul(dAt.T, dBt.T))):
            assert np.array_equal(got.get(), ref), (M, K, N, name)
        X = rng.standard_normal((M, K)).astype(np.float32); Y = rng.standard_normal((K, N)).astype(np.float32)
        got = nd.matmul(nd.asarray(np.ascontiguousarray(X.T)).T, nd.asarray(Y)).get()
        ref64 = X.astype(np.float64) @ Y.astype(np.float64)
        scale = np.abs(X).astype(np.float64) @ np.abs(Y).astype(np.float64)
        assert (np.abs(got - ref64) <= 2e-6 * scale + 1e-30).all(), (M, K, N)
        assert np.array_equal(nd.matmul(nd.asarray(np.ascontiguousarray(X.T)).T, nd.asarray(Y)).get(), got)      # fixed order


def _three_axis_broadcasts(nd):
    """(B, R, C) op (B, 1, C) / (1, R, 1) / a sliced (B, R, C) view: iteration spaces that keep three axes take the 16-byte
    vector kernel of elementwise.hip (k_ew_axes); bit-identical to NumPy in every loop dtype it serves, both operand orders,
    and the shapes next to its conditions (inner extent not a multiple of 4, misaligned base, small totals) still agree."""
    rng = np.random.default_rng(5)
    for dt in (np.float32, np.float64, np.int32, np.int64):
        def mk(*shape):
            return (rng.standard_normal(shape) * 50).astype(dt)
        for B, R, C in ((16, 96, 128), (16, 96, 132), (16, 96, 130), (3, 5, 8192)):
            x = mk(B, R, C)
            dx = nd.asarray(x)
            for o in (mk(B, 1, C), mk(1, R, 1), mk(B, 1, 1), mk(B, R, 1), mk(R, 1)):
                do = nd.asarray(o)
                for name in ("multiply", "subtract", "maximum"):
                    np.testing.assert_array_equal(getattr(nd, name)(dx, do).get(), getattr(np, name)(x, o), err_msg=f"{dt.__name__} {name} {x.shape} {o.shape}")
                    np.testing.assert_array_equal(getattr(nd, name)(do, dx).get(), getattr(np, name)(o, x), err_msg=f"{dt.__name__} {name} {o.shape} {x.shape}")
            np.testing.assert_array_equal(nd.multiply(dx, 3).get(), x * 3)
    # strided operands reach the kernel as views, not copies
    base = rng.standard_normal((8, 70, 264)).astype(np.float32)
    db = nd.asarray(base)
    y = rng.standard_normal((8, 1, 256)).astype(np.float32)
    dy = nd.asarray(y)
    for sl in ((slice(None), slice(2, 66), slice(4, 260)), (slice(None), slice(2, 66), slice(1, 257)), (slice(None), slice(0, 64), slice(0, 256)),
               (slice(None), slice(None, None, -1), slice(8, 264)), (slice(None), slice(0, 64), slice(0, 512, 2))):
        np.testing.assert_array_equal(nd.add(db[sl], dy[:, :, :base[sl].shape[2]]).get(), base[sl] + y[:, :, :base[sl].shape[2]])
        np.testing.assert_array_equal(nd.true_divide(dy[:, :, :base[sl].shape[2]], db[sl]).get(), y[:, :, :base[sl].shape[2]] / base[sl])
        np.testing.assert_array_equal(nd.multiply(db[sl], db[sl]).get(), base[sl] * base[sl])
    # unary calls, comparisons and `where` on the same iteration spaces; four axes
    for sl in ((slice(None), slice(2, 66), slice(4, 260)), (slice(None), slice(2, 66), slice(2, 258)), (slice(None), slice(2, 66), slice(1, 257))):
        np.testing.assert_allclose(nd.exp(db[sl]).get(), np.exp(base[sl]), rtol=1e-6)
        np.testing.assert_array_equal(nd.absolute(db[sl]).get(), np.abs(base[sl]))
        np.testing.assert_array_equal(nd.negative(nd.transpose(db[sl], (1, 0, 2))).get(), -base[sl].transpose(1, 0, 2))
        np.testing.assert_array_equal(nd.greater(db[sl], dy).get(), base[sl] > y)
        m = base[sl] > 0
        dm = nd.asarray(m)
        np.testing.assert_array_equal(nd.where(dm, db[sl], dy).get(), np.where(m, base[sl], y))
        np.testing.assert_array_equal(nd.where(dm, dy, 0.0).get(), np.where(m, y, np.float32(0)))
        np.testing.assert_array_equal(nd.where(nd.greater(dy, 0), db[sl], 1.5).get(), np.where(y > 0, base[sl], np.float32(1.5)))
        np.testing.assert_array_equal(nd.logical_and(dm, nd.greater(dy, 0)).get(), m & (y > 0))
        np.testing.assert_array_equal(nd.isnan(db[sl]).get(), np.isnan(base[sl]))
    x4 = rng.standard_normal((6, 10, 12, 128)).astype(np.float32)
    d4 = nd.asarray(x4)
    for shp in ((1, 10, 1, 128), (6, 1, 12, 1), (1, 10, 1, 1), (6, 1, 1, 128), (6, 10, 1, 1), (10, 1, 128)):
        o = rng.standard_normal(shp).astype(np.float32)
        np.testing.assert_array_equal(nd.multiply(d4, nd.asarray(o)).get(), x4 * o)
        np.testing.assert_array_equal(nd.subtract(nd.asarray(o), d4).get(), o - x4)
        np.testing.assert_array_equal(nd.where(nd.less(d4, nd.asarray(o)), d4, nd.asarray(o)).get(), np.where(x4 < o, x4, o))
    big4 = rng.standard_normal((6, 11, 13, 136)).astype(np.float64)
    np.testing.assert_array_equal(nd.sqrt(nd.absolute(nd.asarray(big4)[:, 1:, 1:, 4:132])).get(), np.sqrt(np.abs(big4[:, 1:, 1:, 4:132])))
    np.testing.assert_array_equal(nd.add(nd.asarray(big4)[:, 1:, 1:, 2:130], nd.asarray(big4)[:, :10, :12, 6:134]).get(), big4[:, 1:, 1:, 2:130] + big4[:, :10, :12, 6:134])
    # fused chains (lazy mode) on the same iteration spaces: the interpreter's vector kernel for three / four axes
    was = nd.lazy_enabled() if hasattr(nd, "lazy_enabled") else None
    nd.set_lazy(True)
    try:
        g = rng.standard_normal((8, 1, 256)).astype(np.float32)
        h = rng.standard_normal((1, 64, 1)).astype(np.float32)
        dg, dh = nd.asarray(g), nd.asarray(h)
        for sl in ((slice(None), slice(2, 66), slice(4, 260)), (slice(None), slice(2, 66), slice(1, 257)), (slice(None), slice(0, 64), slice(0, 256))):
            xs = base[sl]
            got = nd.materialize(nd.power(nd.add(nd.multiply(db[sl], dg), dh), 2)).get()
            np.testing.assert_allclose(got, (xs * g + h) ** 2, rtol=1e-6, atol=1e-6)
            got = nd.materialize(nd.where(nd.greater(db[sl], dg), nd.subtract(db[sl], dh), nd.negative(dg))).get()
            np.testing.assert_array_equal(got, np.where(xs > g, xs - h, -g))
            got = nd.materialize(nd.greater(nd.add(db[sl], dh), dg)).get()
            np.testing.assert_array_equal(got, (xs + h) > g)
        o4 = rng.standard_normal((1, 10, 1, 128)).astype(np.float32)
        p4 = rng.standard_normal((6, 1, 12, 1)).astype(np.float32)
        got = nd.materialize(nd.multiply(nd.subtract(d4, nd.asarray(o4)), nd.asarray(p4))).get()
        np.testing.assert_array_equal(got, (x4 - o4) * p4)
        i3 = rng.integers(-50, 50, (8, 64, 256)).astype(np.int64)
        j3 = rng.integers(-50, 50, (8, 1, 256)).astype(np.int32)
        got = nd.materialize(nd.add(nd.multiply(nd.asarray(i3), nd.asarray(j3)), 3)).get()
        np.testing.assert_array_equal(got, i3 * j3 + 3)
    finally:
        nd.set_lazy(False if was is None else was)
    # the broadcast operand is itself a broadcast view; the two big operands are different views of one base
    np.testing.assert_array_equal(nd.add(db[:, :64, :256], nd.broadcast_to(dy[:1], (8, 64, 256))).get(), base[:, :64, :256] + y[:1])
    np.testing.assert_array_equal(nd.subtract(db[:, :64, :256], db[:, 6:70, 8:264]).get(), base[:, :64, :256] - base[:, 6:70, 8:264])


def test_three_axis_broadcasts_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _three_axis_broadcasts(nd)


@pytest.mark.gpu
def test_three_axis_broadcasts_vector_kernel_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _three_axis_broadcasts(nd)


def _linear_layer_on_3d_input(nd):
    """(B, M, K) @ (K, N): the batch of a dense `a` folds into the rows of ONE product (md_build_gemm); views whose batch is not laid
    out as more rows, a broadcast `a`, and every dtype keep the batched route. Same values either way."""
    rng = np.random.default_rng(6)
    for dt in (np.float32, np.float64, np.int32, np.int64):
        def mk(*shape):
            return (rng.standard_normal(shape) * 4).astype(dt)
        w = mk(40, 24)
        for a in (mk(6, 5, 40), mk(6, 1, 40), mk(1, 5, 40), mk(3, 2, 5, 40), mk(6, 10, 40)[:, ::2], mk(6, 5, 48)[:, :, 4:44], mk(5, 6, 40).transpose(1, 0, 2)):
            ref = np.matmul(a.astype(np.float64), w.astype(np.float64))
            da = nd.asarray(np.ascontiguousarray(a)) if a.base is None else None
            if da is None:
                da = nd.asarray(a)
            got = nd.matmul(da, nd.asarray(w)).get()
            assert got.shape == ref.shape and got.dtype == dt
            if np.dtype(dt).kind == "f":
                np.testing.assert_allclose(got, ref, rtol=2e-5 if dt is np.float32 else 1e-12, atol=1e-4 if dt is np.float32 else 1e-10)
            else:
                np.testing.assert_array_equal(got, ref.astype(dt))
        # device-side views: a strided batch (no fold), a sliced K (fold: rows keep one stride)
        a = mk(6, 10, 48)
        da = nd.asarray(a)
        for sl in ((slice(None), slice(None, None, 2), slice(4, 44)), (slice(None), slice(None), slice(0, 40)), (slice(1, 5), slice(None), slice(8, 48))):
            got, ref = nd.matmul(da[sl], nd.asarray(w)).get(), np.matmul(a[sl].astype(np.float64), w.astype(np.float64))
            if np.dtype(dt).kind == "f":
                np.testing.assert_allclose(got, ref, rtol=2e-5 if dt is np.float32 else 1e-12, atol=1e-4 if dt is np.float32 else 1e-10)
            else:
                np.testing.assert_array_equal(got, ref.astype(dt))
        # out= : a dense 3-D result
        a = mk(4, 8, 40)
        out = nd.asarray(np.zeros((4, 8, 24), dt))
        nd.matmul(nd.asarray(a), nd.asarray(w), out=out)
        ref = np.matmul(a.astype(np.float64), w.astype(np.float64))
        assert np.allclose(out.get(), ref, rtol=2e-5, atol=1e-4)


def test_linear_layer_on_3d_input_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _linear_layer_on_3d_input(nd)


@pytest.mark.gpu
def test_linear_layer_on_3d_input_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _linear_layer_on_3d_input(nd)
    rng = np.random.default_rng(7)
    a, w = rng.standard_normal((16, 96, 512)).astype(np.float32), rng.standard_normal((512, 256)).astype(np.float32)
    got = nd.matmul(nd.asarray(a), nd.asarray(w)).get()
    np.testing.assert_allclose(got, np.matmul(a.astype(np.float64), w.astype(np.float64)), rtol=1e-4, atol=1e-3)


def _middle_axis_narrow_inner(nd):
    """Reductions over the middle axis of (8, n, 8) / (3, n, 100) / (40, n, 40): few outputs in short contiguous runs (block per output),
    a wave-wide kept axis (column kernels), many outputs — same values whichever kernel the dispatch picks."""
    rng = np.random.default_rng(8)
    for shp in ((8, 6250, 8), (3, 3000, 100), (40, 700, 40), (2, 5000, 64), (1, 9000, 63)):
        for dt in (np.float32, np.int64, np.bool_):
            a = (rng.integers(-4, 5, shp)).astype(dt)
            d = nd.asarray(a)
            for name in ("sum", "max", "min", "any"):
                got, ref = getattr(nd, name)(d, axis=1).get(), getattr(np, name)(a, axis=1)
                assert got.dtype == ref.dtype and np.array_equal(got, ref), (shp, dt, name)
            np.testing.assert_array_equal(nd.sum(d, axis=1, keepdims=True).get(), np.sum(a, axis=1, keepdims=True))


def test_middle_axis_narrow_inner_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _middle_axis_narrow_inner(nd)


@pytest.mark.gpu
def test_middle_axis_narrow_inner_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _middle_axis_narrow_inner(nd)


def _narrow_broadcasts(nd):
    """Storage-only dtypes under broadcasts (narrow.hip k_nw_binary_axes): row vectors, (B, 1, C), (1, R, 1), four axes, sliced views with
    aligned and misaligned offsets, comparisons (bool results), scalars — bit-identical to NumPy."""
    rng = np.random.default_rng(9)
    for dt in (np.float16, np.int8, np.uint8, np.int16, np.uint16, np.uint32, np.uint64):
        def mk(*shape):
            if np.dtype(dt).kind == "f":
                return (rng.standard_normal(shape) * 4).astype(dt)
            return rng.integers(0, 100, shape).astype(dt)
        for shape, others in (((96, 256), [(256,), (96, 1), (1, 256)]), ((6, 40, 128), [(6, 1, 128), (1, 40, 1), (40, 128), (6, 40, 1)]),
                              ((3, 4, 20, 64), [(1, 4, 1, 64), (3, 1, 20, 1)]), ((70, 250), [(250,)]), ((64, 264), [(264,)])):
            x = mk(*shape)
            dx = nd.asarray(x)
            for oshape in others:
                o = mk(*oshape)
                do = nd.asarray(o)
                with np.errstate(all="ignore"):
                    for name in ("add", "multiply", "maximum", "less", "not_equal"):
                        want, got = getattr(np, name)(x, o), getattr(nd, name)(dx, do).get()
                        assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (dt, name, shape, oshape)
                        want, got = getattr(np, name)(o, x), getattr(nd, name)(do, dx).get()
                        assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (dt, name, oshape, shape)
        base = mk(6, 44, 160)
        db = nd.asarray(base)
        y = mk(6, 1, 128)
        dy = nd.asarray(y)
        for sl in ((slice(None), slice(2, 42), slice(16, 144)), (slice(None), slice(2, 42), slice(1, 129)), (slice(None), slice(0, 40), slice(8, 136)),
                   (slice(None), slice(None, None, -1), slice(32, 160))):
            ys = y if base[sl].shape[1] == 40 else y
            with np.errstate(all="ignore"):
                np.testing.assert_array_equal(nd.add(db[sl], dy).get(), base[sl] + ys)
                np.testing.assert_array_equal(nd.greater_equal(dy, db[sl]).get(), ys >= base[sl])
                np.testing.assert_array_equal(nd.subtract(db[sl], db[sl]).get(), base[sl] - base[sl])


def test_narrow_broadcasts_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    _narrow_broadcasts(nd)


@pytest.mark.gpu
def test_narrow_broadcasts_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    _narrow_broadcasts(nd)
