"""std along one axis through mdhip_var (csrc/moments.hip; reference call sites minidiff/backend/numpy.py:57,
minidiff/ops/definitions.py:209-221): one read of the array for rows, two for columns, instead of the five array passes of
NumPy's _var composed from the other entry points. Which forms take the kernel, that they agree with np.std, and that every
other form still goes through the composition. Runs on whichever library the session is bound to."""
import numpy as np
import pytest

from minidiff_amd import ndarray as nd


def _tol(dtype):
    return 2e-6 if dtype == np.float32 else 1e-13


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rows_and_columns_take_the_kernel_and_match_numpy(lib, dtype):
    rng = np.random.default_rng(57)
    prev = nd.set_lazy(False)
    try:
        shapes_axes = [((7, 8), 1), ((5, 3, 64), 2), ((33, 2048), -1), ((9, 2052), 1), ((3, 5000), 1), ((300, 20000), 1), ((256, 70000), -1),
                       ((64, 256), 0), ((1000, 512), 0), ((129, 260), 0), ((4096,), 0), ((4096,), None)]
        for shape, axis in shapes_axes:
            h = (rng.standard_normal(shape) * 3 + 10).astype(dtype)          # a mean well away from 0: E[x^2] - mean^2 would lose digits
            d = nd.asarray(h)
            for ddof in (0, 1):
                for keep in (False, True):
                    n = h.size if axis is None else h.shape[axis]
                    fused = nd._std_fused(d, axis, None, ddof, keep, n)
                    assert fused is not None, (shape, axis)
                    exp = np.std(h, axis=axis, ddof=ddof, keepdims=keep)
                    got = nd.std(d, axis=axis, ddof=ddof, keepdims=keep)
                    assert got.shape == exp.shape and got.dtype == exp.dtype, (shape, axis, keep)
                    np.testing.assert_allclose(got.get(), exp, rtol=_tol(dtype) * 8, atol=0)
                    np.testing.assert_array_equal(fused.get(), got.get())
    finally:
        nd.set_lazy(prev)


def test_other_forms_are_composed_as_before(lib):
    rng = np.random.default_rng(58)
    prev = nd.set_lazy(False)
    try:
        h = rng.standard_normal((6, 10, 12)).astype(np.float32)
        d = nd.asarray(h)
        cases = [
            (d, 1, None, 0), (d, (0, 2), None, 0), (d, None, None, 0),     # middle axis, two axes, all axes of a 3-D array
            (d[:, :, ::2], 2, None, 0),                                     # not contiguous
            (nd.asarray(h[:, :, :10].copy()), 2, None, 0),                  # rows of 10 floats: not a multiple of 16 B
            (nd.asarray(h.astype(np.int32)), 2, None, 0),                   # integers: NumPy answers in float64
            (d, 2, np.float64, 0),                                          # a different accumulation dtype
            (nd.asarray(h[0, :, :4].copy()), 0, None, 0),                   # a narrow, short column form
            (d, 2, None, 12), (d, 2, None, 13),                             # degenerate counts: NumPy's inf / nan
            (nd.asarray(rng.standard_normal((2, 20000)).astype(np.float32)), 1, None, 0),   # two long rows: one block each would idle the chip
        ]
        with np.errstate(all="ignore"):
            for arr, axis, dt, ddof in cases:
                n = arr.size if axis is None else int(np.prod([arr.shape[a] for a in (axis if isinstance(axis, tuple) else (axis,))]))
                assert nd._std_fused(arr, axis, dt, ddof, False, n) is None, (arr.shape, axis, dt, ddof)
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    exp = np.std(arr.get(), axis=axis, dtype=dt, ddof=ddof)
                    got = nd.std(arr, axis=axis, dtype=dt, ddof=ddof)
                assert got.dtype == exp.dtype and got.shape == exp.shape
                np.testing.assert_allclose(got.get(), exp, rtol=2e-5, equal_nan=True)
        with pytest.raises(np.exceptions.AxisError):
            nd.std(d, axis=3)
    finally:
        nd.set_lazy(prev)


def test_tape_std_forward_and_backward_against_the_oracle(engines):
    hip, ora = engines
    x = np.random.default_rng(59).standard_normal((64, 256)).astype(np.float64) + 2.0
    out = {}
    for name, md in (("hip", hip), ("ora", ora)):
        t = md.Tensor(x, allow_grad=True)
        y = md.std(t, axis=(0,))       # (axis 0: the reference's vjp broadcasts the un-kept mean against x)
        md.sum(y * y).backward()
        out[name] = (y.as_numpy(), t.grad.as_numpy())
    np.testing.assert_allclose(out["hip"][0], out["ora"][0], rtol=1e-12)
    np.testing.assert_allclose(out["hip"][1], out["ora"][1], rtol=1e-10, atol=1e-13)


@pytest.mark.gpu
def test_full_size_std_gpu(on_gpu):
    assert on_gpu
    rng = np.random.default_rng(60)
    h = (rng.standard_normal((8192, 4096)) * 2 + 5).astype(np.float32)
    d = nd.asarray(h)
    prev = nd.set_lazy(False)
    try:
        for axis in (1, 0):
            got = nd.std(d, axis=axis).get()
            exp = np.std(h.astype(np.float64), axis=axis)
            assert np.abs(got - exp).max() / exp.max() < 2e-6
            assert np.array_equal(nd.std(d, axis=axis).get(), got)      # bit-identical run to run
    finally:
        nd.set_lazy(prev)
