"""hipGraph capture / replay of a sweep (include/mdhip.h "hipGraph capture / replay",
minidiff_amd/graph.py). Numerical replay needs the device; the CPU double only checks
that the call sequence is accepted and that a failed capture leaves the library usable."""
import numpy as np
import pytest

from minidiff_amd import ndarray as nd, workloads
from minidiff_amd.graph import CapturedSweep


def _cfg3_expected(xh, yh):
    s = np.sin(xh) * yh
    return (np.float32(2) * s * yh * np.cos(xh)).astype(np.float32), (np.float32(2) * s * np.sin(xh)).astype(np.float32)


def test_capture_on_cpu_double_fails_cleanly(engines, on_gpu):
    if on_gpu:
        pytest.skip("CPU double behaviour")
    hip, _ = engines
    state, step = workloads.make_cfg3(hip, n=1000)
    with pytest.raises(RuntimeError, match="cannot replay"):
        CapturedSweep(step)
    out = step()  # the library is still usable, and a new capture may begin
    assert out["x"].grad.as_numpy().shape == (1000,)
    with pytest.raises(RuntimeError, match="cannot replay"):
        CapturedSweep(step)


@pytest.mark.gpu
@pytest.mark.parametrize("lazy", [False, True])
def test_replay_recomputes_from_current_inputs(engines, lazy):
    hip, _ = engines
    nd.set_lazy(lazy)
    try:
        n = 1 << 18
        state, step = workloads.make_cfg3(hip, n=n)
        x, y = state["x"], state["y"]
        sweep = CapturedSweep(step)
        out = sweep.replay()
        gx, gy = _cfg3_expected(x.as_numpy(), y.as_numpy())
        np.testing.assert_allclose(out["x"].grad.as_numpy(), gx, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(out["y"].grad.as_numpy(), gy, rtol=2e-5, atol=1e-6)

        # unrelated allocations after the capture must not land on the graph's reserved blocks
        junk = [nd.full((n,), float(i)) for i in range(24)]
        # new inputs go INTO the captured arrays; the replay recomputes everything from them
        rng = np.random.default_rng(11)
        x2 = rng.standard_normal(n, dtype=np.float32)
        y2 = rng.standard_normal(n, dtype=np.float32)
        x._data[...] = nd.asarray(x2)
        y._data[...] = nd.asarray(y2)
        out = sweep.replay()
        gx, gy = _cfg3_expected(x2, y2)
        np.testing.assert_allclose(out["x"].grad.as_numpy(), gx, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(out["y"].grad.as_numpy(), gy, rtol=2e-5, atol=1e-6)
        s = np.sin(x2.astype(np.float64)) * y2
        np.testing.assert_allclose(float(out["out"].as_numpy()), float((s * s).sum()), rtol=1e-5)
        assert float(np.asarray(junk[3])[0]) == 3.0
        sweep.close()
    finally:
        nd.set_lazy(False)


@pytest.mark.gpu
def test_replay_matmul_sweep_matches_eager(engines):
    hip, _ = engines
    state, step = workloads.make_cfg4(hip, batch=256, d_in=192, d_out=160)
    eager = step()
    w_ref, b_ref = eager["W"].grad.as_numpy().copy(), eager["b"].grad.as_numpy().copy()
    sweep = CapturedSweep(step)
    for _ in range(3):
        out = sweep.replay()
    np.testing.assert_array_equal(out["W"].grad.as_numpy(), w_ref)  # same kernels, same order: bit-identical
    np.testing.assert_array_equal(out["b"].grad.as_numpy(), b_ref)
    sweep.close()
    again = step()
    np.testing.assert_array_equal(again["W"].grad.as_numpy(), w_ref)


@pytest.mark.gpu
def test_synchronising_call_aborts_capture(engines):
    hip, _ = engines
    state, step = workloads.make_cfg3(hip, n=4096)

    def bad():
        out = step()
        out["x"].grad.as_numpy()  # D2H needs a stream synchronisation: not capturable
        return out

    with pytest.raises(RuntimeError):
        CapturedSweep(bad)
    out = step()  # capture was aborted; eager work goes on
    gx, _ = _cfg3_expected(state["x"].as_numpy(), state["y"].as_numpy())
    np.testing.assert_allclose(out["x"].grad.as_numpy(), gx, rtol=2e-5, atol=1e-6)


# ---- reuse_graph: tape-side memoisation (reference minidiff/caching.py) and the hipGraph cache keyed by it ----
def test_reuse_graph_memoises_traversal_by_structure(engines):
    """Inside `md.reuse_graph()` two graphs of one structure share ONE cached traversal (keyed by the root's
    structural hash), a different structure gets its own, and gradients equal the un-cached ones."""
    for md in engines:
        def sweep(kind):
            x = md.Tensor(np.arange(6.0).reshape(2, 3) / 4, allow_grad=True)
            y = md.Tensor(np.full((2, 3), 2.0), allow_grad=True)
            f = md.sum((md.sin(x) * y) ** 2 + x * y) if kind == 0 else md.sum(md.cos(x) * y)
            f.backward()
            return np.asarray(x.grad.as_numpy()).copy(), np.asarray(y.grad.as_numpy()).copy(), f

        plain = sweep(0)
        assert plain[2].op_node.op_ids is None          # no structural ids outside the context
        with md.reuse_graph():
            a = sweep(0)
            h_a = md.last_root_hash
            b = sweep(0)
            assert md.last_root_hash == h_a
            c = sweep(1)
            h_c = md.last_root_hash
            assert h_c != h_a
            assert len(md.backward_paths_for_root(a[2].op_node)) == len(md.backward_paths_for_root(b[2].op_node))
            assert md.backward_paths_for_root(a[2].op_node) is md.backward_paths_for_root(b[2].op_node)   # memoised
        for got in (a, b):
            np.testing.assert_array_equal(got[0], plain[0])
            np.testing.assert_array_equal(got[1], plain[1])
        with pytest.raises(ValueError):
            md.backward_paths_for_root(a[2].op_node)     # "Not currently preserving graph" (caching.py:33-34)


def test_sweep_signature_separates_what_the_structural_hash_merges(engines):
    """the structural hash maps every leaf and scalar to -1 (as the reference's does), so
    sum(sin(a)*sin(a)) and sum(sin(a)*sin(b)) share it — a hipGraph captured for one would replay for the other and
    leave b.grad untouched. graph.SweepCache keys on (hash, tape signature): the signature tells them apart, and also
    shapes, scalar constants, keyword arguments and shared intermediates."""
    from minidiff_amd.graph import SweepCache
    for md in engines:
        a = md.Tensor(np.arange(6.0).reshape(2, 3) / 4, allow_grad=True)
        b = md.Tensor(np.arange(6.0).reshape(2, 3) / 8, allow_grad=True)
        c = md.Tensor(np.arange(4.0) / 8, allow_grad=True)
        keys = {}
        cache = SweepCache(md)

        def key_of(name, fn):
            with md.reuse_graph():
                fn().backward()
                keys[name] = cache._key()

        key_of("aa", lambda: md.sum(md.sin(a) * md.sin(a)))
        key_of("ab", lambda: md.sum(md.sin(a) * md.sin(b)))
        key_of("aa2", lambda: md.sum(md.sin(a) * md.sin(a)))
        def shared():
            s = md.sin(a)
            return md.sum(s * s)                        # ONE sin node used twice: another kernel sequence than sin(a)*sin(a)
        key_of("shared", shared)
        key_of("pow2", lambda: md.sum(a ** 2))
        key_of("pow3", lambda: md.sum(a ** 3))
        key_of("shape", lambda: md.sum(md.sin(c) * md.sin(c)))
        key_of("axis0", lambda: md.sum(md.sum(a * b, axis=(0,))))
        key_of("axis1", lambda: md.sum(md.sum(a * b, axis=(1,))))
        assert keys["aa"][0] == keys["ab"][0]           # the reference-style hash cannot tell them apart ...
        assert keys["aa"] != keys["ab"]                 # ... the cache key can
        assert keys["aa"] == keys["aa2"]
        assert len({keys[k] for k in ("aa", "ab", "shared", "shape")}) == 4
        assert keys["pow2"] != keys["pow3"] and keys["axis0"] != keys["axis1"]
        cache.close()


def test_sweep_cache_falls_back_to_eager_on_cpu_double(engines, on_gpu):
    if on_gpu:
        pytest.skip("CPU double behaviour")
    from minidiff_amd.graph import SweepCache
    hip, _ = engines
    state, step = workloads.make_cfg3(hip, n=1000)
    ref = step()["x"].grad.as_numpy().copy()
    with SweepCache(hip) as cache:
        for _ in range(4):
            out = cache.run(step)
            np.testing.assert_array_equal(out["x"].grad.as_numpy(), ref)
        assert cache.stats["uncapturable"] == 1 and cache.stats["replayed"] == 0 and cache.stats["eager"] == 4


@pytest.mark.gpu
def test_sweep_cache_replays_and_recaptures_changed_graph(engines):
    """1st run eager, 2nd captured, then replays; when the sweep's structure changes (a flag read by `step`),
    the validation run notices the new structural hash, drops the stale graph, and the new structure is captured."""
    from minidiff_amd.graph import SweepCache
    hip, _ = engines
    n = 1 << 16
    rng = np.random.default_rng(3)
    xh, yh = rng.standard_normal(n, dtype=np.float32), rng.standard_normal(n, dtype=np.float32)
    x, y = hip.Tensor(xh, allow_grad=True), hip.Tensor(yh, allow_grad=True)
    mode = {"kind": 0}

    def step():
        x.grad = None
        y.grad = None
        loss = hip.sum((hip.sin(x) * y) ** 2) if mode["kind"] == 0 else hip.sum(hip.cos(x) * y)
        loss.backward()
        return {"gx": x.grad, "gy": y.grad}   # the RESULT arrays (a replay rewrites them; it cannot rebind x.grad)

    gx0, gy0 = _cfg3_expected(xh, yh)
    with SweepCache(hip, validate_every=3) as cache:
        for i in range(6):
            out = cache.run(step)
            np.testing.assert_allclose(out["gx"].as_numpy(), gx0, rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(out["gy"].as_numpy(), gy0, rtol=2e-5, atol=1e-6)
        assert cache.stats["captured"] == 1 and cache.stats["replayed"] >= 2 and cache.stats["invalidated"] == 0
        # new inputs written INTO the resident tensors are picked up by the replays
        x2 = rng.standard_normal(n, dtype=np.float32)
        x._data[...] = nd.asarray(x2)
        out = cache.run(step)
        np.testing.assert_allclose(out["gx"].as_numpy(), _cfg3_expected(x2, yh)[0], rtol=2e-5, atol=1e-6)
        # the graph changes: replays still run the OLD structure until the next validation run...
        mode["kind"] = 1
        seen_new = False
        for i in range(8):
            out = cache.run(step)
            got = out["gx"].as_numpy()
            new = np.allclose(got, -np.sin(x2) * yh, rtol=2e-5, atol=1e-6)
            old = np.allclose(got, _cfg3_expected(x2, yh)[0], rtol=2e-5, atol=1e-6)
            assert new or old
            if seen_new:
                assert new, "after the re-capture every run must follow the new structure"
            seen_new = seen_new or new
        assert seen_new and cache.stats["invalidated"] == 1 and cache.stats["captured"] == 2


def test_sweep_signature_never_copies_arrays_to_the_host(engines, monkeypatch):
    """`_signature` used to repr() every non-Tensor op input — a getitem key holding Tensors / device arrays reached
    DeviceArray.__repr__ -> .get(): one D2H copy and a stream sync per such op in every backward() under reuse_graph (and, inside
    a stream capture, an uncapturable sweep). Array-likes are now identified by (type, shape, dtype, first-sighting index)."""
    from minidiff_amd.graph import SweepCache
    hip, _ = engines
    x = hip.Tensor(np.arange(24.0).reshape(6, 4) / 8, allow_grad=True)
    idx = hip.Tensor(np.array([4, 0, 4, 2]))
    idx2 = hip.Tensor(np.array([1, 1, 3]))
    raw = nd.asarray(np.array([5, 5, 0]))            # a raw DeviceArray inside the key
    gets = {"n": 0}
    plain_get = nd.DeviceArray.get

    def counting_get(self, *a, **kw):
        gets["n"] += 1
        return plain_get(self, *a, **kw)

    cache = SweepCache(hip)
    keys = {}
    monkeypatch.setattr(nd.DeviceArray, "get", counting_get)
    for name, key in (("t", (idx, slice(None))), ("t_again", (idx, slice(None))), ("t_other_shape", (idx2, slice(None))),
                      ("raw", (raw, slice(1, 3))), ("raw_slice2", (raw, slice(0, 3)))):
        with hip.reuse_graph():
            x.grad = None
            hip.sum(x[key] * 2.0).backward()
            keys[name] = cache._key()
    monkeypatch.undo()
    assert gets["n"] == 0, "the sweep signature must not read device arrays back"
    assert keys["t"] == keys["t_again"]
    assert len({keys["t"], keys["t_other_shape"], keys["raw"], keys["raw_slice2"]}) == 4
    # small HOST constants are still told apart by value (a replay bakes their upload in), without a repr of the device side
    with hip.reuse_graph():
        hip.sum(x * np.array([1.0, 2.0, 3.0, 4.0])).backward(); k1 = cache._key()
        hip.sum(x * np.array([1.0, 2.0, 3.0, 5.0])).backward(); k2 = cache._key()
    assert k1 != k2
    cache.close()


@pytest.mark.gpu
def test_tensor_indexed_getitem_sweep_is_captured(engines):
    """A sweep with a gather by a resident index tensor and its scatter-add backward (definitions.py:186-189) under SweepCache on
    the GPU. Eagerly mdhip_gather / mdhip_scatter read their bounds verdict back (NumPy raises IndexError at the call); inside a
    capture they cannot — the kernels skip out-of-range positions, a scatter whose bounds pass found one writes nothing, and the
    verdict waits in a pinned word for the next synchronisation (csrc/index.hip, md_sticky_check). So the sweep IS capturable, the
    signature adds no device-to-host copy, and replays follow the index tensor's CURRENT contents."""
    from minidiff_amd.graph import SweepCache
    hip, _ = engines
    rng = np.random.default_rng(11)
    xh = rng.standard_normal((64, 32)).astype(np.float32)
    ih = rng.integers(0, 64, 200)
    x, idx = hip.Tensor(xh, allow_grad=True), hip.Tensor(ih)

    def step():
        x.grad = None
        hip.sum(x[(idx, slice(None))] * 3.0).backward()
        return {"gx": x.grad}

    def expect(i):
        e = np.zeros_like(xh)
        np.add.at(e, i, np.float32(3.0))
        return e

    with SweepCache(hip, validate_every=0) as cache:
        for _ in range(5):
            out = cache.run(step)
            np.testing.assert_array_equal(out["gx"].as_numpy(), expect(ih))
        assert cache.stats["uncapturable"] == 0 and cache.stats["captured"] == 1 and cache.stats["replayed"] >= 2, cache.stats
        # other indices in the SAME buffer: the replayed kernels read them
        ih2 = rng.integers(-64, 64, 200)
        idx._data[...] = nd.asarray(ih2)
        out = cache.run(step)
        np.testing.assert_array_equal(out["gx"].as_numpy(), expect(ih2))
        # an index out of range at replay time: nothing faults, the error arrives with the next synchronisation — once
        bad = ih2.copy(); bad[17] = 64
        idx._data[...] = nd.asarray(bad)
        out = cache.run(step)
        with pytest.raises(IndexError, match="replayed graph"):
            out["gx"].as_numpy()
        idx._data[...] = nd.asarray(ih)
        out = cache.run(step)
        np.testing.assert_array_equal(out["gx"].as_numpy(), expect(ih))


@pytest.mark.gpu
def test_captured_gather_scatter_defer_their_bounds_verdict(lib, on_gpu):
    """Library level: gathers / scatters of every flavour recorded into one graph (rows, elements, along-axis; SET and ADD, float and
    integer), replayed with good and then with bad indices. A scatter that met a bad index leaves its destination untouched."""
    assert on_gpu
    import ctypes as C
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(5)
    table = rng.standard_normal((512, 64)).astype(np.float32)
    rows = rng.integers(0, 512, 6000)                     # > 4096 rows: the sorted row path
    few = rng.integers(0, 512, 40)                        # the serial path
    cols = rng.integers(0, 64, (512, 3))
    t, r, f, c = nd.asarray(table), nd.asarray(rows), nd.asarray(few), nd.asarray(cols)
    acc, acc_i, small = nd.zeros((512, 64), np.float32), nd.zeros((512, 64), np.int64), nd.zeros((512, 64), np.float32)
    upd = nd.asarray(np.ones((6000, 64), np.float32))
    bins = rng.integers(0, 10, 6000)                      # element granularity, ~600 contributions per destination: the sorted path
    binv = rng.standard_normal(6000).astype(np.float32)
    hist, b_, bv_ = nd.zeros((16,), np.float32), nd.asarray(bins), nd.asarray(binv)
    outs = {}

    def body():
        outs["g"] = t[r]                                  # gather rows
        outs["e"] = t[f, 3]                               # gather elements
        outs["a"] = nd.take_along_axis(t, c, 1)
        nd.index_add(acc, r, upd)                         # float rows, duplicates: sort + apply
        nd.index_add(acc_i, r, 1)                         # integer: atomics
        small[f] = 2.0                                    # serial SET
        nd.index_add(hist, b_, bv_)                       # sort by destination + one serial pass each (no host rounds)

    nd._lib().sync()
    h = C.c_void_p()
    nd._lib().graph_begin()
    try:
        body()
    finally:
        nd._lib().graph_end(C.byref(h))
    try:
        nd._lib().graph_launch(h)
        nd._lib().sync()
        e_acc = np.zeros((512, 64), np.float32); np.add.at(e_acc, rows, 1.0)
        assert np.array_equal(outs["g"].get(), table[rows]) and np.array_equal(outs["e"].get(), table[few, 3])
        assert np.array_equal(outs["a"].get(), np.take_along_axis(table, cols, 1))
        assert np.array_equal(acc.get(), e_acc) and np.array_equal(acc_i.get(), e_acc.astype(np.int64))
        e_small = np.zeros((512, 64), np.float32); e_small[few] = 2.0
        assert np.array_equal(small.get(), e_small)
        e_hist = np.zeros(16, np.float32); np.add.at(e_hist, bins, binv)
        assert np.array_equal(hist.get(), e_hist)
        # bad row index in the scatter-add's plan: the replay raises at the synchronisation and the scatters wrote nothing
        rows_bad = rows.copy(); rows_bad[4321] = 512
        r[...] = nd.asarray(rows_bad)
        nd._lib().graph_launch(h)
        with pytest.raises(IndexError, match="replayed graph"):
            nd._lib().sync()
        assert np.array_equal(acc.get(), e_acc) and np.array_equal(acc_i.get(), e_acc.astype(np.int64))
        nd._lib().sync()                                  # reported once
        r[...] = nd.asarray(rows)
        nd._lib().graph_launch(h)
        nd._lib().sync()
        assert np.array_equal(acc.get(), 2 * e_acc)
    finally:
        nd._lib().graph_destroy(h)


@pytest.mark.gpu
def test_deferred_index_errors_on_request(lib, on_gpu):
    """Option index_defer (MDHIP_INDEX_DEFER=1): eager gathers / scatters skip the read-back of their bounds verdict — a 100-row
    lookup drops from ~47 us to the launch cost — and an out-of-range index is reported by the NEXT synchronisation instead of the call;
    a scatter that met one writes nothing. Off (the default): NumPy's IndexError at the call."""
    assert on_gpu
    from minidiff_amd import ndarray as nd
    rng = np.random.default_rng(2)
    table = rng.standard_normal((100, 16)).astype(np.float32)
    t = nd.asarray(table)
    good, bad = rng.integers(0, 100, 50), np.array([3, 100, 7])
    dgood, dbad = nd.asarray(good), nd.asarray(bad)
    with pytest.raises(IndexError):
        t[dbad]                                                  # default: at the call
    nd._lib().debug_set_option(b"index_defer", 1)
    try:
        assert np.array_equal(t[dgood].get(), table[good])
        acc = nd.zeros((100, 16), np.float32)
        nd.index_add(acc, dgood, t[dgood])
        exp = np.zeros((100, 16), np.float32); np.add.at(exp, good, table[good])
        assert np.array_equal(acc.get(), exp)
        out = t[dbad]                                            # no error here ..
        nd.index_add(acc, dbad, 1.0)                             # .. nor here, and nothing is written
        with pytest.raises(IndexError, match="reported at this synchronisation"):
            nd._lib().sync()
        assert np.array_equal(acc.get(), exp)
        nd._lib().sync()                                         # reported once
        del out
    finally:
        nd._lib().debug_set_option(b"index_defer", 0)
    with pytest.raises(IndexError):
        t[dbad]
