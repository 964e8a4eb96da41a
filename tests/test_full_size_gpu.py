"""BASELINE.json's configs at their FULL sizes on the MI355X, against the NumPy
oracle on the same seeded inputs (gradients norm-wise within 1e-5, north_star),
plus size-independent properties of the matmul path (linearity, transpose
identity, identity operand). Eager and lazy-fusion modes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture
def mode(request, lib):
    from minidiff_amd import ndarray as nd
    prev = nd.set_lazy(request.param == "lazy")
    yield request.param
    nd.set_lazy(prev)


@pytest.mark.parametrize("mode", ["eager", "lazy"], indirect=True)
@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_full_size_gradients_match_oracle(engines, on_gpu, mode, cfg):
    assert on_gpu
    dev, oracle = engines
    from minidiff_amd import workloads
    maker = workloads.MAKERS[cfg]
    s_d, step_d = maker(dev)
    s_o, step_o = maker(oracle)
    out_d = step_d()
    out_o = step_o()
    names = {"cfg2": ("A", "B"), "cfg3": ("x", "y"), "cfg4": ("W", "b"), "cfg5": ("A", "B")}[cfg]
    if cfg == "cfg4":
        _check_cfg4(dev, s_d, s_o, out_d, out_o, mode)
        names = ()
    for n in names:
        g_d, g_o = s_d[n].grad.as_numpy(), s_o[n].grad.as_numpy()
        assert g_d.shape == g_o.shape and g_d.dtype == g_o.dtype, (cfg, n)
        e = _rel(g_d, g_o)
        assert e < 1e-5, (cfg, mode, n, e)
    e = _rel(out_d["out"].as_numpy(), out_o["out"].as_numpy())
    assert e < 1e-5, (cfg, mode, "forward", e)
    # a second sweep must reproduce the first bit-for-bit (deterministic kernels, no atomics)
    n0 = names[0] if names else "W"
    first = s_d[n0].grad.as_numpy().copy()
    step_d()
    assert np.array_equal(first, s_d[n0].grad.as_numpy()), (cfg, mode)


def _check_cfg4(dev, s_d, s_o, out_d, out_o, mode):
    """relu is discontinuous: among 33.5 M pre-activations a few lie within rounding of 0,
    and a different summation order in the GEMM flips their mask bit, which moves W.grad by
    a whole row of X. So: (1) the masks must agree wherever |z| is not tiny, (2) the
    gradients must equal the closed form X^T m / sum_rows m for the DEVICE's own mask m."""
    X, W, b = s_d["X"].as_numpy(), s_d["W"].as_numpy(), s_d["b"].as_numpy()
    with dev.no_grad():
        z_d = (s_d["X"] @ s_d["W"] + s_d["b"]).as_numpy()
    z_o = X @ W + b
    assert _rel(z_d, z_o) < 1e-5
    m_d, m_o = z_d > 0, z_o > 0
    disagree = m_d != m_o
    assert disagree.mean() < 1e-5
    assert np.abs(z_o[disagree]).max(initial=0.0) < 1e-4 * np.abs(z_o).max()
    exp_W = X.astype(np.float64).T @ m_d.astype(np.float64)
    exp_b = m_d.sum(axis=0, dtype=np.float64)
    assert _rel(s_d["W"].grad.as_numpy(), exp_W) < 1e-5, mode
    assert _rel(s_d["b"].grad.as_numpy(), exp_b) < 1e-6, mode
    assert _rel(out_d["out"].as_numpy(), np.where(m_d, z_d, 0).sum(dtype=np.float64)) < 1e-5
    # (3) ... and they are tied to the ORACLE's gradients at the full 8192 x 4096 size: a flipped mask[i, j]
    # moves column j of W.grad by row i of X and b.grad[j] by 1, so with the flips COUNTED per column
    #     |W.grad_dev - W.grad_oracle|[k, j] <= 1e-5 max|W.grad| + flips[j] max|X|      |b.grad_dev - b.grad_oracle|[j] <= 1e-6 max|b.grad| + flips[j]
    flips = disagree.sum(axis=0).astype(np.float64)
    gW_d, gW_o = s_d["W"].grad.as_numpy().astype(np.float64), s_o["W"].grad.as_numpy().astype(np.float64)
    gb_d, gb_o = s_d["b"].grad.as_numpy().astype(np.float64), s_o["b"].grad.as_numpy().astype(np.float64)
    bound_W = 1e-5 * np.abs(gW_o).max() + flips[None, :] * float(np.abs(X).max())
    bound_b = 1e-6 * np.abs(gb_o).max() + flips
    assert (np.abs(gW_d - gW_o) <= bound_W).all(), (mode, int(flips.sum()), float((np.abs(gW_d - gW_o) - bound_W).max()))
    assert (np.abs(gb_d - gb_o) <= bound_b).all(), (mode, int(flips.sum()))
    # columns without a flip (all but a few tens of the 4096) agree outright
    clean = flips == 0
    assert clean.sum() >= 0.97 * clean.size
    assert _rel(gW_d[:, clean], gW_o[:, clean]) < 1e-5 and _rel(gb_d[clean], gb_o[clean]) < 1e-6, mode


def test_matmul_properties_4096(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(7)
    n = 4096
    A = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
    B1 = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
    B2 = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
    C1 = nd.matmul(A, B1)
    # linearity in the right operand
    lhs = nd.matmul(A, nd.add(B1, B2)).get()
    rhs = nd.add(C1, nd.matmul(A, B2)).get()
    assert _rel(lhs, rhs) < 1e-5
    # (A B)^T = B^T A^T through the strided-view loaders (TT path)
    assert _rel(nd.matmul(B1.T, A.T).get(), C1.get().T) < 1e-6
    # identity operand reproduces B exactly (products with 0 / 1 are exact, k-ordered fma chain)
    I = nd.asarray(np.eye(n, dtype=np.float32))
    assert np.array_equal(nd.matmul(I, B1).get(), B1.get())
    assert np.array_equal(nd.matmul(B1, I).get(), B1.get())
    # sampled rows against float64 on the host
    Ah, Bh, Ch = A.get(), B1.get(), C1.get()
    rows = rng.integers(0, n, 6)
    ref = Ah[rows].astype(np.float64) @ Bh.astype(np.float64)
    assert _rel(Ch[rows], ref) < 1e-5  # fp32 fma chain of length 4096 vs float64: ~3e-6 (north_star bound 1e-5)


@pytest.mark.gpu
def test_more_than_2_to_31_elements_gpu(lib, on_gpu):
    """64-bit indexing end to end: fills, streaming kernels, reductions, arg-reductions, views, gathers and
    scatters on arrays with more than 2**31 elements (8.6 GB of float32; the card has 288 GB)."""
    assert on_gpu
    import numpy as np
    from minidiff_amd import ndarray as nd
    n = (1 << 31) + 4099
    b = nd.ones((n,), dtype=np.bool_)
    assert int(np.asarray(nd.sum(b))) == n                      # int64 accumulation: exact
    x = nd.full((n,), 0.5, dtype=np.float32)
    mark = (1 << 31) + 7
    x[mark:mark + 1] = 3.0
    x[5:6] = -2.0
    y = nd.add(nd.multiply(x, 2.0), 1.0)                        # two streaming kernels over 8.6 GB each way
    assert float(np.asarray(y[mark])) == 7.0 and float(np.asarray(y[n - 1])) == 2.0 and float(np.asarray(y[5])) == -3.0
    assert int(np.asarray(nd.argmax(y))) == mark and int(np.asarray(nd.argmin(y))) == 5
    assert float(np.asarray(nd.max(y))) == 7.0
    s = float(np.asarray(nd.sum(y, dtype=np.float64)))
    assert s == 2.0 * (n - 2) + 7.0 - 3.0
    tail = y[(1 << 31):]                                        # a view starting beyond 2**31
    assert tail.shape == (4099,) and float(np.asarray(tail[7])) == 7.0
    idx = nd.asarray(np.array([5, mark, n - 1, 0], dtype=np.int64))
    assert np.array_equal(np.asarray(y[idx]), np.array([-3.0, 7.0, 2.0, 2.0], dtype=np.float32))
    nd.index_add(y, idx, nd.asarray(np.array([1.0, 1.0, 1.0, 1.0], dtype=np.float32)))
    assert np.array_equal(np.asarray(y[idx]), np.array([-2.0, 8.0, 3.0, 3.0], dtype=np.float32))
    m = nd.reshape(y[: 4096 * 524289], (524289, 4096))          # 2-D, 2**31 + 4096 elements
    col = np.asarray(nd.sum(m, axis=0, dtype=np.float64))
    assert col.shape == (4096,) and col[1] == 2.0 * 524289 and col[7] == 2.0 * 524288 + 8.0
    row = np.asarray(nd.sum(m, axis=1, dtype=np.float64))
    assert row.shape == (524289,) and row[1] == 2.0 * 4096 and row[524288] == 2.0 * 4095 + 8.0


@pytest.mark.gpu
def test_matmul_with_more_than_2_to_31_outputs_gpu(lib, on_gpu):
    assert on_gpu
    import numpy as np
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(12)
    M, K, N = 65536, 64, 32800                     # 2.15e9 outputs (8.6 GB of float32)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    c = nd.matmul(nd.asarray(a), nd.asarray(b))
    assert c.shape == (M, N)
    rows = np.array([0, 1, 32767, 32768, 65535])
    got = np.asarray(c[nd.asarray(rows)])
    ref = a[rows].astype(np.float64) @ b.astype(np.float64)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6
    cols = np.array([0, 31, 32799])
    got = np.asarray(c[:, nd.asarray(cols)])
    ref = a.astype(np.float64) @ b[:, cols].astype(np.float64)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6


@pytest.mark.parametrize("world", [2, 4, 8])
def test_cfg4_rank_shard_shapes(engines, on_gpu, world):
    """The per-rank GEMM shapes of the batch-sharded cfg4 sweep (at 8 ranks (1024 x 4096 x 4096) NN and
    (4096 x 1024 x 4096) TN): one rank's shard on the device against the oracle's closed form for the device's own
    mask, and the shard gradients of all ranks summed on the host against the full-batch gradient."""
    assert on_gpu
    dev, _ = engines
    from minidiff_amd import workloads
    rank = world - 1   # the last shard (its rows end the batch)
    s_d, step_d = workloads.make_cfg4(dev, rank=rank, world=world)
    step_d()
    X, W, b = s_d["X"].as_numpy(), s_d["W"].as_numpy(), s_d["b"].as_numpy()
    assert X.shape == (8192 // world, 4096)
    with dev.no_grad():
        z = (s_d["X"] @ s_d["W"] + s_d["b"]).as_numpy()
    assert _rel(z, X @ W + b) < 1e-5
    m = z > 0
    assert _rel(s_d["W"].grad.as_numpy(), X.astype(np.float64).T @ m.astype(np.float64)) < 1e-5
    assert _rel(s_d["b"].grad.as_numpy(), m.sum(axis=0, dtype=np.float64)) < 1e-6


def test_weight_gradient_in_row_panels_is_bit_identical(engines, on_gpu):
    """dp.GradSync produces W.grad = X^T @ G in row panels inside the all-reduce bucket (one GEMM per panel, each
    followed by its collective on the second stream). Every output element is the same k-ordered fma chain, so at
    world size 1 the panelled, all-reduced gradient must equal the plain sweep's bit for bit — at the rank-shard
    shape of 8 ranks and at the full batch."""
    assert on_gpu
    from minidiff_amd import dp, workloads
    hip, _ = engines
    comm = dp.RcclComm(0, 1)
    try:
        for kw in ({"batch": 1024}, {"batch": 8192}):
            st, step = workloads.make_cfg4(hip, **kw)
            step()
            ref = {k: st[k].grad.as_numpy().copy() for k in ("W", "b")}
            for panels in (4, 8, 1):
                sync = dp.GradSync(hip, st["params"], comm, force=True, panels=panels)
                for _ in range(2):
                    step()
                    sync()
                assert sync.overlapped == 2 and sync.panel_collectives == (2 * panels if panels > 1 else 0)
                for k in ("W", "b"):
                    np.testing.assert_array_equal(st[k].grad.as_numpy(), ref[k])
                sync.close()
    finally:
        comm.close()


@pytest.mark.parametrize("glds", ["1", "0"])
@pytest.mark.parametrize("tile", range(8))
def test_every_gemm_tile_config_exact_on_integers(lib, on_gpu, tile, glds, mdopt):
    """Every tile of the f32 MFMA kernel (gemm.hip's CFG_* list, forced through the option gemm_cfg: mdhip_debug_set_option,
    effective from the next launch), in the three layouts of definitions.py:487-492, whole and ragged shapes, plain and bias+relu
    epilogue kernels: small-integer operands make the f32 fma chain exact, so the results must EQUAL NumPy's."""
    assert on_gpu
    from minidiff_amd import ndarray as nd
    mdopt("gemm_cfg", tile)
    mdopt("gemm_glds", glds)     # direct-to-LDS kernels (whole aligned tiles) / register-staged kernels
    rng = np.random.default_rng(100 + tile)
    prev = nd.set_lazy(False)
    try:
        for (M, K, N) in ((512, 128, 256), (300, 70, 200), (256, 2048, 128), (256, 96, 256), (256, 32, 512),
                          (300, 72, 200), (260, 100, 132), (516, 36, 260), (131, 64, 257)):   # (ragged with 4-multiples: zero-filled DMA edges)
            A = rng.integers(-4, 5, (M, K)).astype(np.float32)
            B = rng.integers(-4, 5, (K, N)).astype(np.float32)
            ref = A.astype(np.float64) @ B
            dA, dB = nd.asarray(A), nd.asarray(B)
            dAt, dBt = nd.asarray(np.ascontiguousarray(A.T)), nd.asarray(np.ascontiguousarray(B.T))
            for tag, a, b in (("NN", dA, dB), ("NT", dA, dBt.T), ("TN", dAt.T, dB), ("TT", dAt.T, dBt.T)):
                assert np.array_equal(nd.matmul(a, b).get(), ref), (tile, tag, M, K, N)
        # epilogue kernel of the same tile (lazy mode recognises sum(where(X@W+b > 0, X@W+b, 0)))
        nd.set_lazy(True)
        M, K, N = 512, 64, 256
        X = rng.integers(-3, 4, (M, K)).astype(np.float32)
        W = rng.integers(-3, 4, (K, N)).astype(np.float32)
        b = (rng.integers(-3, 4, N) + 0.5).astype(np.float32)     # (never exactly 0 after the add)
        s0 = nd.FUSION_STATS["gemm_epilogue"]
        z = nd.add(nd.matmul(nd.asarray(X), nd.asarray(W)), nd.asarray(b))
        m = nd.greater(z, 0)
        loss = nd.sum(nd.where(m, z, 0))
        zr = X.astype(np.float64) @ W + b
        assert nd.FUSION_STATS["gemm_epilogue"] - s0 == 1, tile
        assert np.array_equal(m.get(), zr > 0), tile
        assert abs(float(loss.get()) - np.where(zr > 0, zr, 0).sum()) <= 1e-6 * np.abs(zr).sum(), tile
    finally:
        nd.set_lazy(prev)


def test_direct_to_lds_gemm_random_aligned_shapes(lib, on_gpu):
    """Random whole-tile and ragged (multiples of 4) shapes, batches, aligned sub-views and transposed operands through the tile picker (no forced config):
    integer-valued operands, so every product must EQUAL NumPy's; a float case per shape within 2e-6."""
    assert on_gpu
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(2026)
    prev = nd.set_lazy(False)
    try:
        for case in range(60):
            M, N = (int(rng.choice([128, 256, 384, 512, 768, 1024, 132, 260, 500, 1000])) for _ in range(2))
            K = int(rng.choice([32, 64, 96, 160, 256, 480, 1024, 36, 100, 420]))
            batch = int(rng.choice([0, 0, 0, 2, 3]))
            lay = str(rng.choice(["NN", "NT", "TN", "TT"]))
            lead = (batch,) if batch else ()
            # operands cut out of larger arrays at 16-B aligned offsets (row stride stays a multiple of 4 floats)
            pad_r, pad_c = int(rng.choice([0, 4, 8])), int(rng.choice([0, 4, 12]))
            def make(rows, cols, transposed):
                r, c = (cols, rows) if transposed else (rows, cols)
                big = rng.integers(-3, 4, lead + (r + pad_r, c + pad_c)).astype(np.float32)
                dev = nd.asarray(big)
                sl = (slice(None),) * len(lead) + (slice(pad_r, pad_r + r), slice(pad_c, pad_c + c))
                h, d = big[sl], dev[sl]
                return (np.swapaxes(h, -1, -2), nd.swapaxes(d, -1, -2)) if transposed else (h, d)
            (ha, da), (hb, db) = make(M, K, lay[0] == "T"), make(K, N, lay[1] == "T")
            got = nd.matmul(da, db).get()
            assert np.array_equal(got, np.matmul(ha.astype(np.float64), hb.astype(np.float64))), (case, M, K, N, batch, lay, pad_r, pad_c)
            fa, fb = rng.standard_normal((M, K), dtype=np.float32), rng.standard_normal((K, N), dtype=np.float32)
            ref = fa.astype(np.float64) @ fb
            assert _rel(nd.matmul(nd.asarray(fa), nd.asarray(fb)).get(), ref) < 2e-6, (case, M, K, N)
    finally:
        nd.set_lazy(prev)


def test_tt_products_run_as_the_swapped_nn_product(lib, on_gpu, mdopt):
    """x.T @ y.T of two row-major arrays ("TT") has no kernel of its own: C^T = y x is the NN product of the two storages, run on
    the direct-to-LDS NN kernels with C addressed through swapped strides and stored as 16-B vectors along rows (gemm.hip,
    HipExec::gemm). Integer-valued operands: every product must EQUAL NumPy's and the register-staged TT kernel's
    (option gemm_tt_swap = 0); whole tiles of every size class, three-buffer grids, batches, ragged sizes."""
    assert on_gpu
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(77)
    prev = nd.set_lazy(False)
    try:
        shapes = [(1024, 512, 768, 0), (2048, 256, 2048, 0), (4096, 128, 4096, 0), (512, 384, 256, 3), (1000, 260, 516, 0), (130, 64, 258, 2),
                  (1024, 1024, 1024, 0)]
        for (M, K, N, batch) in shapes:
            lead = (batch,) if batch else ()
            xa = rng.integers(-3, 4, lead + (K, M)).astype(np.float32)      # A = xa^T
            xb = rng.integers(-3, 4, lead + (N, K)).astype(np.float32)      # B = xb^T
            da, db = nd.asarray(xa), nd.asarray(xb)
            ref = np.matmul(np.swapaxes(xa, -1, -2).astype(np.float64), np.swapaxes(xb, -1, -2).astype(np.float64))
            mdopt("gemm_tt_swap", 1)
            got = nd.matmul(nd.swapaxes(da, -1, -2), nd.swapaxes(db, -1, -2))
            assert got.shape == ref.shape and got.is_c_contiguous
            assert np.array_equal(got.get(), ref), (M, K, N, batch)
            mdopt("gemm_tt_swap", 0)
            old = nd.matmul(nd.swapaxes(da, -1, -2), nd.swapaxes(db, -1, -2))
            assert np.array_equal(old.get(), ref), (M, K, N, batch)
    finally:
        nd.set_lazy(prev)


def test_misaligned_operands_of_large_products_are_repacked(lib, on_gpu, mdopt):
    """Odd leading dimensions (x.T of a matrix with an odd column count) and views that start off a 16-byte boundary: large
    products copy such an operand once into an aligned, row-padded buffer and take the direct-to-LDS kernels (gemm.hip,
    HipExec::gemm); the padding must never reach C. Integer-valued operands: the products must EQUAL NumPy's, in every layout,
    with the repack on and off."""
    assert on_gpu
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(4097)
    prev = nd.set_lazy(False)
    try:
        for (M, K, N) in ((1027, 512, 516), (513, 260, 1031), (771, 256, 771)):
            for lay in ("NN", "NT", "TN", "TT"):
                for off in (0, 1, 3):
                    def make(rows, cols, transposed):
                        r, c = (cols, rows) if transposed else (rows, cols)
                        big = rng.integers(-3, 4, (r, c + off)).astype(np.float32)     # odd row stride and / or an offset start
                        h, d = big[:, off:], nd.asarray(big)[:, off:]
                        return (h.T, d.T) if transposed else (h, d)
                    (ha, da), (hb, db) = make(M, K, lay[0] == "T"), make(K, N, lay[1] == "T")
                    ref = np.matmul(ha.astype(np.float64), hb.astype(np.float64))
                    for flag in ("1", "0"):
                        mdopt("gemm_repack", flag)      # (effective from the next launch: the "0" leg really runs the edge kernels)
                        got = nd.matmul(da, db).get()
                        assert np.array_equal(got, ref), (M, K, N, lay, off, flag)
        mdopt("gemm_repack", 1)
        # the shape DESIGN §9.1 quotes, TN with an odd M: result against float64
        M, K, N = 4097, 4096, 4100
        a = rng.standard_normal((K, M), dtype=np.float32)
        b = rng.standard_normal((K, N), dtype=np.float32)
        got = nd.matmul(nd.asarray(a).T, nd.asarray(b)).get()
        ref = a.T.astype(np.float64)[:64] @ b.astype(np.float64)
        assert _rel(got[:64], ref) < 5e-6
        ref = a.T.astype(np.float64)[-33:] @ b.astype(np.float64)
        assert _rel(got[-33:], ref) < 5e-6
    finally:
        nd.set_lazy(prev)


def test_peeled_ragged_products(lib, on_gpu, mdopt):
    """A product a few rows / columns past a multiple of 256 runs as an aligned main block on the whole-tile kernels plus a bottom
    and a right strip (gemm.hip, launch_mfma_peeled). Forced here (option gemm_peel = 2) on every layout, with thin and fat strips,
    batches, odd leading dimensions (repacked operands) and K that the peel refuses: integer-valued operands, the products must
    EQUAL NumPy's and the single-launch result (gemm_peel = 0)."""
    assert on_gpu
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(256)
    prev = nd.set_lazy(False)
    try:
        shapes = [(257, 64, 260, 0), (513, 96, 512, 0), (512, 128, 516, 0), (640, 64, 300, 2), (1025, 512, 1028, 0), (384, 100, 260, 0), (300, 64, 257, 0)]
        for (M, K, N, batch) in shapes:
            lead = (batch,) if batch else ()
            for lay in ("NN", "NT", "TN", "TT"):
                def make(rows, cols, transposed):
                    r, c = (cols, rows) if transposed else (rows, cols)
                    big = rng.integers(-3, 4, lead + (r, c)).astype(np.float32)
                    dev = nd.asarray(big)
                    return (np.swapaxes(big, -1, -2), nd.swapaxes(dev, -1, -2)) if transposed else (big, dev)
                (ha, da), (hb, db) = make(M, K, lay[0] == "T"), make(K, N, lay[1] == "T")
                ref = np.matmul(ha.astype(np.float64), hb.astype(np.float64))
                for mode in ("2", "1", "0"):
                    mdopt("gemm_peel", mode)
                    assert np.array_equal(nd.matmul(da, db).get(), ref), (M, K, N, batch, lay, mode)
        # the model's own choice at the shape the peel exists for (TN, odd M: repacked A, one bottom row, four right columns)
        mdopt("gemm_peel", 1)
        M, K, N = 4097, 1024, 4100
        a = rng.integers(-2, 3, (K, M)).astype(np.float32)
        b = rng.integers(-2, 3, (K, N)).astype(np.float32)
        got = nd.matmul(nd.asarray(a).T, nd.asarray(b)).get()
        ref = a.T.astype(np.float64) @ b.astype(np.float64)
        assert np.array_equal(got, ref)
    finally:
        nd.set_lazy(prev)


def test_gemm_whole_tile_shapes_with_unusual_strides(lib, on_gpu):
    """Whole-tile shapes whose operands are flipped (negative strides), broadcast (stride 0) or every-other-row views: the
    direct-to-LDS launchers must either take them correctly or leave them to the register-staged kernel — exact on integers."""
    assert on_gpu
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(77)
    M = K = N = 256
    A = rng.integers(-3, 4, (2 * M, K)).astype(np.float32)
    B = rng.integers(-3, 4, (K, 2 * N)).astype(np.float32)
    dA, dB = nd.asarray(A), nd.asarray(B)
    prev = nd.set_lazy(False)
    try:
        cases = [
            (dA[:M], dB[:, :N], A[:M], B[:, :N]),                                  # plain sub-views
            (nd.flip(dA[:M], 0), dB[:, :N], A[:M][::-1], B[:, :N]),                # negative row stride on A
            (dA[:M], nd.flip(dB[:, :N], 1), A[:M], B[:, :N][:, ::-1]),             # negative column stride on B
            (dA[::2], dB[:, ::2], A[::2], B[:, ::2]),                              # every other row / column
            (nd.broadcast_to(dA[:1], (M, K)), dB[:, :N], np.broadcast_to(A[:1], (M, K)), B[:, :N]),   # stride-0 rows
            (nd.flip(dA[:M], 0).T, nd.flip(dA[:M], 1), A[:M][::-1].T, A[:M][:, ::-1]),                # TN with flipped operands
        ]
        for i, (a, b, ha, hb) in enumerate(cases):
            assert np.array_equal(nd.matmul(a, b).get(), ha.astype(np.float64) @ hb.astype(np.float64)), i
    finally:
        nd.set_lazy(prev)
