"""Opt-in device random numbers (csrc/md_rng.h, rng.hip, ndarray.device_rng; reference aliases backend/numpy.py:131-137).
The stream is pinned by tests/golden/rng_stream.npz (written by the CPU test double, which compiles the same generator): both
targets must replay it — bit for bit where no libm is involved. An independent NumPy restatement of Philox4x32-10 checks the
generator itself; distribution checks run at sizes where 6 sigma is a comfortable bound."""
import os

import numpy as np
import pytest

from golden.make_rng_golden import draw

gpu = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "rng_stream.npz"))


def _twin(on_gpu, want_gpu):
    if want_gpu != on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    return nd


def _stream(on_gpu, want_gpu):
    nd = _twin(on_gpu, want_gpu)
    got = draw(nd)
    assert set(got) == set(GOLD.files)
    for k, v in got.items():
        exp = GOLD[k]
        assert v.dtype == exp.dtype and v.shape == exp.shape, k
        if k.startswith("normal"):
            assert np.abs(v - exp).max() <= (2e-6 if v.dtype == np.float32 else 1e-12), k
        else:
            assert np.array_equal(v, exp), k


def test_stream_matches_fixture_cpu(lib, on_gpu): _stream(on_gpu, False)
@gpu
def test_stream_matches_fixture_gpu(lib, on_gpu): _stream(on_gpu, True)


def _philox_np(counter, seed):
    """Philox4x32-10 (Salmon et al. 2011) on uint64 counters: the four output words per counter, key = seed."""
    m0, m1, w0, w1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85
    mask = np.uint64(0xFFFFFFFF)
    counter = np.asarray(counter, np.uint64)
    c0, c1 = counter & mask, counter >> np.uint64(32)
    c2, c3 = np.full_like(c0, 0x6D646870), np.zeros_like(c0)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = m0 * c0, m1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + w0) & 0xFFFFFFFF, (k1 + w1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint64)


def _generator(on_gpu, want_gpu):
    nd = _twin(on_gpu, want_gpu)
    seed = 0x1234_5678_9ABC_DEF0
    prev = nd.device_rng(True, seed=seed)
    try:
        n = 1000
        u = nd.random_uniform((n,), np.float32).get()                 # words 0..n-1 of blocks 0..
        words = _philox_np(np.arange((n + 3) // 4), seed).reshape(-1)[:n]
        assert np.array_equal(u, ((words >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)))
        # the next call starts at the next whole block
        u2 = nd.random_uniform((5,), np.float64).get()
        w2 = _philox_np(np.arange((n + 3) // 4, (n + 3) // 4 + 3), seed).reshape(-1)
        exp = ((((w2[0:10:2] << np.uint64(32)) | w2[1:10:2]) >> np.uint64(11)).astype(np.float64)) / 9007199254740992.0
        assert np.array_equal(u2, exp)
    finally:
        nd.device_rng(prev)


def test_generator_is_philox_cpu(lib, on_gpu): _generator(on_gpu, False)
@gpu
def test_generator_is_philox_gpu(lib, on_gpu): _generator(on_gpu, True)


def _distributions(on_gpu, want_gpu):
    nd = _twin(on_gpu, want_gpu)
    prev = nd.device_rng(True, seed=7)
    try:
        n = 400_000
        for dt in (np.float32, np.float64):
            u = nd.random_uniform((n,), dt).get()
            assert u.dtype == dt and u.min() >= 0.0 and u.max() < 1.0
            assert abs(u.mean() - 0.5) < 6 * 0.2887 / np.sqrt(n) and abs(u.var() - 1 / 12) < 6 * 0.0745 / np.sqrt(n)
            hist = np.histogram(u, bins=20, range=(0, 1))[0]
            assert (((hist - n / 20) ** 2) / (n / 20)).sum() < 60          # chi-square, 19 dof (mean 19, sd 6.2)
            z = nd.random_normal((n,), dt).get()
            assert z.dtype == dt and np.isfinite(z).all()
            assert abs(z.mean()) < 6 / np.sqrt(n) and abs(z.var() - 1.0) < 6 * np.sqrt(2.0 / n)
            assert abs((np.abs(z) > 2.0).mean() - 0.0455) < 6 * np.sqrt(0.0455 * 0.9545 / n)
            assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 6 / np.sqrt(n)   # neighbours uncorrelated
        k = nd.random_integers(-3, 4, (n,)).get()
        assert k.dtype == np.int64 and k.min() == -3 and k.max() == 3
        cnt = np.bincount(k + 3, minlength=7)
        assert (((cnt - n / 7) ** 2) / (n / 7)).sum() < 40                 # 6 dof
        big = nd.random_integers(0, 1 << 53, (1000,)).get()
        assert big.min() >= 0 and big.max() < (1 << 53) and big.max() > (1 << 50)
        for nn, p in ((1, 0.3), (12, 0.5), (200, 0.9)):
            b = nd.random_binomial(nn, p, (n // 4,)).get()
            m = n // 4
            assert b.min() >= 0 and b.max() <= nn
            assert abs(b.mean() - nn * p) < 6 * np.sqrt(nn * p * (1 - p) / m)
            assert abs(b.var() - nn * p * (1 - p)) < 0.05 * nn * p * (1 - p) + 1e-3
        assert not nd.random_binomial(9, 0.0, (100,)).get().any() and (nd.random_binomial(9, 1.0, (100,)).get() == 9).all()
        with pytest.raises(ValueError):
            nd.random_binomial(257, 0.5, (4,))
        with pytest.raises(ValueError):
            nd.random_integers(3, 3, (4,))
        # permutations: every index once, different each call, reproducible per seed
        p1 = nd.random_permutation(100_000).get()
        p2 = nd.random_permutation(100_000).get()
        assert np.array_equal(np.sort(p1), np.arange(100_000)) and np.array_equal(np.sort(p2), np.arange(100_000))
        assert (p1 != p2).mean() > 0.99 and (p1 == np.arange(100_000)).mean() < 0.001
        assert abs(np.corrcoef(p1, np.arange(100_000))[0, 1]) < 0.02
        nd.device_rng(True, seed=7)
        again = nd.random_uniform((5,), np.float32).get()
        nd.device_rng(True, seed=7)
        assert np.array_equal(again, nd.random_uniform((5,), np.float32).get())
        nd.device_rng(True, seed=8)
        assert not np.array_equal(again, nd.random_uniform((5,), np.float32).get())
        assert nd.random_permutation(0).shape == (0,) and nd.random_uniform((0, 3)).shape == (0, 3)
    finally:
        nd.device_rng(prev)


def test_distributions_cpu(lib, on_gpu): _distributions(on_gpu, False)
@gpu
def test_distributions_gpu(lib, on_gpu): _distributions(on_gpu, True)


def _backend_table(on_gpu, want_gpu):
    """The reference-facing functions (hip_backend) in both modes: device generator on -> device arrays of NumPy's shapes and
    dtypes with nothing drawn on the host; off -> NumPy's own stream for np.random.seed (what the reference gives)."""
    nd = _twin(on_gpu, want_gpu)
    from minidiff_amd.hip_backend import HipBackendTable as T
    prev = nd.device_rng(False)
    try:
        np.random.seed(5)
        host_first = T.randn(3, 4).get()
        np.random.seed(5)
        assert np.array_equal(host_first, np.random.randn(3, 4))            # default: NumPy's numbers for the seed
        nd.device_rng(True, seed=99)
        state = np.random.get_state()[1].copy()
        r = T.rand(6, 5); z = T.randn(2, 3, 4); k = T.randint(10, size=(7,)); k2 = T.randint(-2, 3, (4, 4), np.int32)
        b = T.binomial(1, 0.4, (8, 8)); perm = T.permutation(50)
        assert (r.shape, r.dtype) == ((6, 5), np.float64) and (z.shape, z.dtype) == ((2, 3, 4), np.float64)
        assert (k.shape, k.dtype) == ((7,), np.int64) and 0 <= int(k.get().min()) and int(k.get().max()) < 10
        assert k2.dtype == np.int32 and set(np.unique(k2.get())) <= {-2, -1, 0, 1, 2}
        for lo, hi in ((0, 2 ** 31 + 1), (-2 ** 31 - 1, 0)):      # NumPy's bounds check for the requested dtype: no silent wrap-around
            with pytest.raises(ValueError, match="out of bounds"):
                T.randint(lo, hi, (3,), np.int32)
        assert b.dtype == np.int64 and set(np.unique(b.get())) <= {0, 1}
        assert np.array_equal(np.sort(perm.get()), np.arange(50))
        x = nd.asarray(np.arange(60, dtype=np.float32).reshape(20, 3))
        px = T.permutation(x).get()
        assert px.shape == (20, 3) and np.array_equal(np.sort(px[:, 0]), np.arange(0, 60, 3)) and np.array_equal(px[:, 1] - px[:, 0], np.ones(20))
        y = nd.asarray(np.arange(60, dtype=np.float32).reshape(20, 3))
        T.shuffle(y)
        yh = y.get()
        assert np.array_equal(np.sort(yh[:, 0]), np.arange(0, 60, 3)) and not np.array_equal(yh[:, 0], np.arange(0, 60, 3))
        c = T.choice(10, size=(4, 2), replace=False).get()
        assert c.shape == (4, 2) and len(np.unique(c)) == 8 and c.min() >= 0 and c.max() < 10
        pool = nd.asarray(np.array([2.5, 3.5, 4.5], dtype=np.float64))
        cc = T.choice(pool, size=(50,)).get()
        assert set(np.unique(cc)) <= {2.5, 3.5, 4.5} and len(np.unique(cc)) == 3
        with pytest.raises(ValueError):
            T.choice(3, size=5, replace=False)
        assert np.array_equal(state, np.random.get_state()[1])               # NumPy's global stream was never touched
        # forms the device generator does not cover still work (host draw + upload)
        assert T.binomial(1000, 0.5, (3,)).shape == (3,) and T.choice(4, size=3, p=[0.1, 0.2, 0.3, 0.4]).shape == (3,)
    finally:
        nd.device_rng(prev)


def test_backend_table_cpu(lib, on_gpu): _backend_table(on_gpu, False)
@gpu
def test_backend_table_gpu(lib, on_gpu): _backend_table(on_gpu, True)


def _npy_io(on_gpu, want_gpu, tmp_path):
    """save / load (numpy.py:129-130): NumPy's file format; a path is memory-mapped and uploaded from the mapping, an open file
    object and an .npz member take NumPy's plain reader."""
    nd = _twin(on_gpu, want_gpu)
    from minidiff_amd.hip_backend import HipBackendTable as T
    rng = np.random.default_rng(3)
    for arr in (rng.standard_normal((37, 5)).astype(np.float32), rng.integers(-9, 9, (4, 3, 2)), rng.random((6,)) < 0.5, np.float64(2.5)):
        f = str(tmp_path / "a.npy")
        T.save(f, nd.asarray(arr))
        assert np.array_equal(np.load(f), arr)                       # the file is a plain .npy
        back = T.load(f)
        assert isinstance(back, nd.DeviceArray) and back.dtype == np.asarray(arr).dtype and np.array_equal(back.get(), arr)
        with open(f, "rb") as fh:
            assert np.array_equal(T.load(fh).get(), arr)
    np.save(str(tmp_path / "t.npy"), np.asfortranarray(rng.standard_normal((5, 7))))
    assert np.array_equal(T.load(str(tmp_path / "t.npy")).get(), np.load(str(tmp_path / "t.npy")))   # Fortran-ordered payload


def test_npy_io_cpu(lib, on_gpu, tmp_path): _npy_io(on_gpu, False, tmp_path)
@gpu
def test_npy_io_gpu(lib, on_gpu, tmp_path): _npy_io(on_gpu, True, tmp_path)
