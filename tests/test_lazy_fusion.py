"""Opt-in lazy fusion (minidiff_amd/lazy.py + mdhip_vm_eval / mdhip_vm_reduce):
fused results equal the eager ones (same per-element functors), pending
expressions survive in-place writes to their leaves, programs that outgrow the
interpreter are split, and the fused launches really replace the eager ones."""
import os

import numpy as np
import pytest

gpu = pytest.mark.gpu


@pytest.fixture
def lazy_nd(lib):
    from minidiff_amd import ndarray as nd
    prev = nd.set_lazy(True)
    yield nd
    nd.set_lazy(prev)


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() / scale <= tol, np.abs(a - b).max() / scale


def _chain(nd, on_gpu, want_gpu):
    if want_gpu != on_gpu:
        pytest.skip("other twin")
    rng = np.random.default_rng(0)
    x = rng.standard_normal((64, 256)).astype(np.float32)
    y = rng.standard_normal((64, 256)).astype(np.float32)
    b = rng.standard_normal((256,)).astype(np.float32)
    dx, dy, db = nd.asarray(x), nd.asarray(y), nd.asarray(b)
    before = dict(nd.FUSION_STATS)
    e = nd.multiply(nd.power(nd.multiply(nd.sin(dx), dy), 2), nd.cos(dx))
    e = nd.where(nd.greater(nd.add(e, db), 0), e, 0.5)
    assert e._expr is not None and e._buf is None, "chain should still be pending"
    exp = np.where((np.sin(x) * y) ** 2 * np.cos(x) + b > 0, (np.sin(x) * y) ** 2 * np.cos(x), np.float32(0.5))
    tot = nd.sum(e)                       # fused full reduce
    cols = nd.sum(e, axis=0)              # fused reduce-to-shape
    mx = nd.max(e)
    assert nd.FUSION_STATS["vm_reduce"] - before["vm_reduce"] == 3
    assert e._buf is None, "reductions must not materialise the operand"
    _close(tot.get(), exp.sum(dtype=np.float64), 1e-5)
    _close(cols.get(), exp.sum(axis=0, dtype=np.float64), 1e-5)
    assert mx.get() == exp.max()
    got = e.get()                         # one vm_eval
    assert nd.FUSION_STATS["vm_eval"] - before["vm_eval"] == 1
    assert got.dtype == np.float32
    _close(got, exp, 2e-6)
    # bool-valued program
    m = nd.logical_and(nd.greater(dx, 0), nd.less(dy, 0.5))
    assert m.dtype == np.bool_ and m._expr is not None
    assert np.array_equal(m.get(), (x > 0) & (y < 0.5))
    # strided leaves take the generic interpreter kernel
    t = nd.add(nd.sin(dx.T), dy.T)
    _close(t.get(), np.sin(x.T) + y.T, 2e-6)
    # float64 program
    x64 = rng.standard_normal((33, 7))
    d64 = nd.asarray(x64)
    _close(nd.sum(nd.exp(nd.multiply(d64, 0.5))).get(), np.exp(x64 * 0.5).sum(), 1e-13)
    # integer loops are never fused
    xi = nd.asarray(np.arange(12))
    r = nd.add(xi, 3)
    assert r._expr is None and np.array_equal(r.get(), np.arange(12) + 3)


def test_chain_cpu(lazy_nd, on_gpu): _chain(lazy_nd, on_gpu, False)
@gpu
def test_chain_gpu(lazy_nd, on_gpu): _chain(lazy_nd, on_gpu, True)


def _hazards(nd, on_gpu, want_gpu):
    if want_gpu != on_gpu:
        pytest.skip("other twin")
    x = np.arange(8, dtype=np.float32)
    dx = nd.asarray(x.copy())
    pending = nd.multiply(dx, 2.0)
    assert pending._expr is not None
    dx += 100.0                           # in-place write to a leaf: the pending value is flushed first
    assert np.array_equal(pending.get(), x * 2)
    assert np.array_equal(dx.get(), x + 100)
    p2 = nd.add(dx, 1.0)
    dx[2:4] = -1.0                        # setitem on the leaf
    assert np.array_equal(p2.get(), x + 101)
    p3 = nd.sin(dx)
    v = dx[1:5]                           # views of a leaf alias it; writes through the view flush too
    v *= 0.0
    assert np.allclose(p3.get(), np.sin(np.where((np.arange(8) >= 2) & (np.arange(8) < 4), -1.0, x + 100)), rtol=1e-6)
    # a program that would exceed the interpreter (depth / length) is split transparently
    acc = nd.asarray(np.ones(16, dtype=np.float32))
    y = acc
    for i in range(120):
        y = nd.add(nd.multiply(y, 1.0), 1.0)
    assert np.array_equal(y.get(), np.full(16, 121.0, dtype=np.float32))
    z = nd.asarray(np.full(4, 1.5, dtype=np.float32))
    w = z
    for i in range(6):                    # tree doubles each step
        w = nd.multiply(w, w)
    assert np.allclose(w.get(), np.float32(1.5) ** 64, rtol=1e-5)
    # views / matmul / gather of a pending array materialise it
    q = nd.exp(nd.asarray(np.eye(4, dtype=np.float32)))
    assert np.allclose(q.T.get(), np.exp(np.eye(4)).T)
    assert np.allclose(nd.matmul(nd.cos(nd.asarray(np.eye(4, dtype=np.float32))), q).get(), np.cos(np.eye(4)) @ np.exp(np.eye(4)), rtol=1e-5)


def test_hazards_cpu(lazy_nd, on_gpu): _hazards(lazy_nd, on_gpu, False)
@gpu
def test_hazards_gpu(lazy_nd, on_gpu): _hazards(lazy_nd, on_gpu, True)


def _sweeps(nd, on_gpu, want_gpu):
    """cfg3 / cfg4 through the tape: lazy == eager gradients, with far fewer launches."""
    if want_gpu != on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import workloads
    from minidiff_amd.hip_backend import HipBackendTable
    from minidiff_amd.tape import build_engine
    md = build_engine(HipBackendTable, "lazy")
    nd.set_lazy(False)
    s_e, step_e = workloads.make_cfg3(md, n=4096)
    step_e()
    gx_e, gy_e = s_e["x"].grad.as_numpy(), s_e["y"].grad.as_numpy()
    c_e, stepc_e = workloads.make_cfg4(md, batch=64, d_in=32, d_out=48)
    stepc_e()
    gw_e, gb_e = c_e["W"].grad.as_numpy(), c_e["b"].grad.as_numpy()
    nd.set_lazy(True)
    before = dict(nd.FUSION_STATS)
    s_l, step_l = workloads.make_cfg3(md, n=4096)
    out = step_l()
    assert nd.FUSION_STATS["vm_reduce"] - before["vm_reduce"] == 1      # sum((sin(x)*y)**2) in one pass
    assert nd.FUSION_STATS["vm_eval_multi"] - before["vm_eval_multi"] == 1  # x.grad and y.grad share one call
    assert nd.FUSION_STATS["vm_eval"] - before["vm_eval"] == 0
    _close(s_l["x"].grad.as_numpy(), gx_e, 1e-6)
    _close(s_l["y"].grad.as_numpy(), gy_e, 1e-6)
    before = dict(nd.FUSION_STATS)
    c_l, stepc_l = workloads.make_cfg4(md, batch=64, d_in=32, d_out=48)
    stepc_l()
    assert nd.FUSION_STATS["vm_reduce"] - before["vm_reduce"] == 2      # loss sum + fused bias-gradient column sum
    _close(c_l["W"].grad.as_numpy(), gw_e, 1e-6)
    _close(c_l["b"].grad.as_numpy(), gb_e, 1e-6)


def _multi(nd, on_gpu, want_gpu):
    """materialize_many: pending results of one shape evaluated together equal their separate
    evaluation (on the device above the compiled-path threshold: ONE kernel, leaves read once)."""
    if want_gpu != on_gpu:
        pytest.skip("other twin")
    rng = np.random.default_rng(5)
    for n, dt in ((1 << 19, np.float32), (1000, np.float32), (300_001, np.float64)):
        xh = rng.standard_normal(n).astype(dt)
        yh = rng.standard_normal(n).astype(dt)
        zh = rng.standard_normal((1,)).astype(dt)
        x, y, z = nd.asarray(xh), nd.asarray(yh), nd.asarray(zh)
        s = nd.multiply(nd.sin(x), y)
        outs = [nd.multiply(nd.multiply(s, 2.0), nd.cos(x)),                 # shares sin(x)*y and the leaves
                nd.add(nd.multiply(s, nd.sin(x)), z),                        # + a broadcast one-element leaf
                nd.where(nd.greater(y, 0.25), nd.exp(x), -1.5),              # constants in every program
                nd.subtract(x, y),
                nd.power(nd.absolute(y), 0.5)]                               # a 5th: left for a single pass
        other_shape = nd.multiply(nd.asarray(xh[:10]), 3.0)
        boolean = nd.greater(x, y)                                           # not a float result: evaluated on its own
        exp = [2.0 * (np.sin(xh) * yh) * np.cos(xh), (np.sin(xh) * yh) * np.sin(xh) + zh,
               np.where(yh > 0.25, np.exp(xh), dt(-1.5)), xh - yh, np.abs(yh) ** dt(0.5)]
        before = dict(nd.FUSION_STATS)
        nd.materialize_many(outs + [other_shape, boolean, outs[0], x])
        assert nd.FUSION_STATS["vm_eval_multi"] - before["vm_eval_multi"] == 1   # four results in one call
        assert nd.FUSION_STATS["vm_eval"] - before["vm_eval"] == 3               # the fifth, the other shape, the bool
        for o in outs + [other_shape, boolean]:
            assert o._expr is None
        tol = 2e-6 if dt is np.float32 else 1e-14
        for o, e in zip(outs, exp):
            assert o.dtype == dt
            _close(np.asarray(o), e.astype(dt), tol)
        assert np.array_equal(np.asarray(boolean), xh > yh)
        _close(np.asarray(other_shape), xh[:10] * dt(3.0), tol)


def test_multi_cpu(lazy_nd, on_gpu): _multi(lazy_nd, on_gpu, False)
@gpu
def test_multi_gpu(lazy_nd, on_gpu): _multi(lazy_nd, on_gpu, True)


def test_sweeps_cpu(lazy_nd, on_gpu): _sweeps(lazy_nd, on_gpu, False)
@gpu
def test_sweeps_gpu(lazy_nd, on_gpu): _sweeps(lazy_nd, on_gpu, True)


@gpu
def test_interpreter_only_gpu(on_gpu):
    """MDHIP_JIT=0 (read once per process): the lazy cases above through the built-in
    interpreter kernels alone — the fallback when hiprtc is unavailable."""
    assert on_gpu
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if os.environ.get("MDHIP_JIT") == "0":
        pytest.skip("already running interpreter-only")
    env = dict(os.environ, MDHIP_JIT="0", MDHIP_LAZY="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_lazy_fusion.py"),
                        os.path.join(here, "test_golden_device.py"), "-m", "gpu", "-q", "--no-header", "-p", "no:cacheprovider",
                        "-k", "not interpreter_only"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


# ---- deferred reduce-to-shape: the column sum of a pending expression is owed until the expression is
# ---- materialised (one pass for both: mdhip_vm_eval_reduce_cols), or computed alone if it is wanted first
def _deferred_cols(nd, on_gpu, want_gpu):
    if want_gpu != on_gpu:
        pytest.skip("other twin")
    rng = np.random.default_rng(5)
    R, C = 640, 1024
    z = rng.standard_normal((R, C)).astype(np.float32)
    b = rng.standard_normal((C,)).astype(np.float32)
    dz, db = nd.asarray(z), nd.asarray(b)
    seed = nd.broadcast_to(nd.asarray(np.float32(1.5)), (R, C))
    exp_g = np.float32(1.5) * ((z + b) > 0)

    def pending():
        return nd.multiply(seed, nd.greater(nd.add(dz, db), 0))

    # (1) expression materialised first: ONE pass gives both
    s0 = dict(nd.FUSION_STATS)
    g = pending()
    cs = nd.sum(g, axis=(0,), keepdims=True)
    assert nd.FUSION_STATS["deferred_cols"] - s0["deferred_cols"] == 1
    assert cs.shape == (1, C) and g._buf is None
    flat = nd.reshape(cs, (C,))                      # a view of an unfilled result must not force it
    assert g._buf is None and nd.FUSION_STATS["vm_reduce"] == s0["vm_reduce"]
    gm = nd.matmul(nd.transpose(dz), g)              # needs g in memory -> eval + column sum in one kernel
    import os
    one_pass = nd.FUSION_STATS["vm_eval_reduce_cols"] - s0["vm_eval_reduce_cols"]
    if os.environ.get("MDHIP_JIT") == "0":   # interpreter only: no one-pass kernel, the two passes run separately
        assert one_pass == 0 and nd.FUSION_STATS["vm_eval"] - s0["vm_eval"] == 1
    else:
        assert one_pass == 1 and nd.FUSION_STATS["vm_eval"] == s0["vm_eval"]
    assert nd.FUSION_STATS["vm_reduce"] == s0["vm_reduce"]
    assert np.array_equal(g.get(), exp_g)
    _close(flat.get(), exp_g.sum(axis=0, dtype=np.float64), 1e-6)
    _close(gm.get(), z.T.astype(np.float64) @ exp_g, 1e-5)

    # (2) the reduction is wanted first: computed alone, the expression stays pending
    s0 = dict(nd.FUSION_STATS)
    g = pending()
    cs = nd.sum(g, axis=0)
    got = cs.get()
    assert nd.FUSION_STATS["vm_reduce"] - s0["vm_reduce"] == 1 and g._buf is None
    _close(got, exp_g.sum(axis=0, dtype=np.float64), 1e-6)
    assert np.array_equal(g.get(), exp_g)
    assert nd.FUSION_STATS["vm_eval_reduce_cols"] == s0["vm_eval_reduce_cols"]

    # (3) result dropped before anything ran: nothing owed any more
    s0 = dict(nd.FUSION_STATS)
    g = pending()
    cs = nd.sum(g, axis=0)
    del cs
    assert np.array_equal(g.get(), exp_g)
    assert nd.FUSION_STATS["vm_eval"] - s0["vm_eval"] == 1 and nd.FUSION_STATS["vm_eval_reduce_cols"] == s0["vm_eval_reduce_cols"]

    # (4) a leaf is overwritten in place while the reduction is owed: values are those at call time
    dz2 = nd.asarray(z.copy())
    g = nd.multiply(seed, nd.greater(nd.add(dz2, db), 0))
    cs = nd.sum(g, axis=0)
    dz2 *= -1.0
    _close(cs.get(), exp_g.sum(axis=0, dtype=np.float64), 1e-6)
    assert np.array_equal(g.get(), exp_g)

    # (5) two reductions owed on one expression (sum and max), then an in-place update of one result
    g = pending()
    cs, cm = nd.sum(g, axis=0), nd.max(g, axis=0)
    cs += 1.0
    _close(cs.get(), exp_g.sum(axis=0, dtype=np.float64) + 1.0, 1e-6)
    nd.materialize(g)
    assert np.array_equal(cm.get(), exp_g.max(axis=0))
    # (6) end-of-sweep flush (workloads._finish -> materialize_many) fills results still owed
    g = pending()
    cs = nd.sum(g, axis=0)
    nd.materialize_many([cs])
    assert cs._buf.task is None
    _close(cs.get(), exp_g.sum(axis=0, dtype=np.float64), 1e-6)


def test_deferred_cols_cpu(lazy_nd, on_gpu): _deferred_cols(lazy_nd, on_gpu, False)
@gpu
def test_deferred_cols_gpu(lazy_nd, on_gpu): _deferred_cols(lazy_nd, on_gpu, True)


def test_failed_one_pass_launch_leaves_the_column_sum_owed(lazy_nd, monkeypatch):
    """the fused eval + column-sum pass claims the owed column sum BEFORE the launch. If the launch then fails
    with anything but the "shape not covered" ValueError (an allocation failure, say), the claimed task must be owed
    again — otherwise its result block stays unwritten and is later read as valid data."""
    nd = lazy_nd
    rng = np.random.default_rng(6)
    z = rng.standard_normal((640, 1024)).astype(np.float32)
    dz = nd.asarray(z)
    g = nd.multiply(nd.greater(dz, 0), 1.5)
    cs = nd.sum(g, axis=0)
    assert g._buf is None and cs._buf.task is not None
    real = nd._lib().vm_eval_reduce_cols

    def boom(*a, **k):
        raise MemoryError("injected: partial rows could not be allocated")

    monkeypatch.setattr(nd._lib(), "vm_eval_reduce_cols", boom)
    try:
        with pytest.raises(MemoryError):
            g.materialize()
    finally:
        monkeypatch.undo()
    assert g._buf is None and cs._buf.task is not None and not cs._buf.task.done   # owed again, nothing looks materialised
    exp = (np.float32(1.5) * (z > 0)).astype(np.float32)
    _close(cs.get(), exp.sum(axis=0, dtype=np.float64), 1e-6)                      # and still computed correctly when needed
    assert np.array_equal(g.get(), exp)
    assert real is not None


def _sincos_hoist(nd, on_gpu, want_gpu):
    """sin(x) and cos(x) of one leaf inside a fused program come from ONE sincos (fusion_jit.inc):
    the values must equal the eager kernels' bit for bit, for small and for huge arguments."""
    if want_gpu != on_gpu:
        pytest.skip("other twin")
    rng = np.random.default_rng(9)
    n = 1 << 19
    x = (rng.standard_normal(n) * np.repeat([1.0, 50.0, 1e4, 1e9], n // 4)).astype(np.float32)
    y = rng.standard_normal(n).astype(np.float32)
    prev = nd.set_lazy(False)
    dx, dy = nd.asarray(x), nd.asarray(y)
    es, ec = nd.sin(dx).get(), nd.cos(dx).get()
    eprod = nd.multiply(nd.multiply(nd.sin(dx), dy), nd.cos(dx)).get()
    nd.set_lazy(True)
    try:
        a = nd.multiply(nd.sin(dx), 1.0)
        b = nd.multiply(nd.cos(dx), 1.0)
        nd.materialize_many([a, b])          # multi-output program: sin and cos of the same leaf
        assert np.array_equal(a.get(), es) and np.array_equal(b.get(), ec)
        p = nd.multiply(nd.multiply(nd.sin(dx), dy), nd.cos(dx))
        assert np.array_equal(p.get(), eprod)
        for dt in (np.float64,):
            x64 = x.astype(dt)
            d64 = nd.asarray(x64)
            q = nd.add(nd.sin(d64), nd.cos(d64))
            _close(q.get(), np.sin(x64) + np.cos(x64), 1e-15)
    finally:
        nd.set_lazy(prev)


def test_sincos_hoist_cpu(lazy_nd, on_gpu): _sincos_hoist(lazy_nd, on_gpu, False)
@gpu
def test_sincos_hoist_gpu(lazy_nd, on_gpu): _sincos_hoist(lazy_nd, on_gpu, True)


# ---- GEMM epilogue fusion: matmul is deferred in lazy mode; sum(where(X@W+b > 0, X@W+b, 0)) runs as ONE GEMM whose
# ---- epilogue adds the bias, accumulates the relu sum and writes the mask the backward pass needs
def _gemm_epilogue(nd, on_gpu, want_gpu, engines):
    if want_gpu != on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import workloads
    rng = np.random.default_rng(21)
    M, K, N = 256, 64, 128
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((K, N)) / 8).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dX, dW, db = nd.asarray(X), nd.asarray(W), nd.asarray(b)
    z_ref = X.astype(np.float64) @ W + b
    s0 = dict(nd.FUSION_STATS)
    p = nd.matmul(dX, dW)
    assert p._expr is not None and p._buf is None, "matmul must be deferred in lazy mode"
    z = nd.add(p, db)
    m = nd.greater(z, 0)
    loss = nd.sum(nd.where(m, z, 0))
    assert nd.FUSION_STATS["gemm_epilogue"] - s0["gemm_epilogue"] == 1
    assert nd.FUSION_STATS["gemm_deferred"] == s0["gemm_deferred"] and nd.FUSION_STATS["vm_reduce"] == s0["vm_reduce"]
    assert m._expr is None and m._buf is not None and p._buf is None      # the mask exists, the product was never written
    _close(loss.get(), np.where(z_ref > 0, z_ref, 0).sum(), 1e-5)
    near = np.abs(z_ref) < 1e-5 * np.abs(z_ref).max()                      # entries within rounding of 0 may flip
    assert np.array_equal(m.get()[~near], (z_ref > 0)[~near])
    # the product is still available to anybody who asks (plain GEMM then)
    _close(p.get(), X.astype(np.float64) @ W, 2e-6)
    assert nd.FUSION_STATS["gemm_deferred"] - s0["gemm_deferred"] == 1
    # commuted operands of the add are recognised too; a shape outside the fused kernel's tiles takes the general path
    s0 = dict(nd.FUSION_STATS)
    z2 = nd.add(db, nd.matmul(dX, dW))
    l2 = nd.sum(nd.where(nd.greater(z2, 0), z2, 0))
    assert nd.FUSION_STATS["gemm_epilogue"] - s0["gemm_epilogue"] == 1
    _close(l2.get(), np.where(z_ref > 0, z_ref, 0).sum(), 1e-5)
    s0 = dict(nd.FUSION_STATS)
    Xo = nd.asarray(X[:100])
    z3 = nd.add(nd.matmul(Xo, dW), db)
    l3 = nd.sum(nd.where(nd.greater(z3, 0), z3, 0))
    assert nd.FUSION_STATS["gemm_epilogue"] == s0["gemm_epilogue"] and nd.FUSION_STATS["gemm_deferred"] - s0["gemm_deferred"] == 1
    _close(l3.get(), np.where(z_ref[:100] > 0, z_ref[:100], 0).sum(), 1e-5)
    # another tail (no relu) is not this pattern
    s0 = dict(nd.FUSION_STATS)
    l4 = nd.sum(nd.multiply(nd.add(nd.matmul(dX, dW), db), 2.0))
    assert nd.FUSION_STATS["gemm_epilogue"] == s0["gemm_epilogue"]
    _close(l4.get(), (2 * z_ref).sum(), 1e-5)
    # a deferred product sees its operands as they were at the call: in-place writes flush it first
    dX2 = nd.asarray(X.copy())
    p2 = nd.matmul(dX2, dW)
    dX2 *= 0.0
    _close(p2.get(), X.astype(np.float64) @ W, 2e-6)
    # ... and an expression that took the pending product as a leaf is flushed before the product is overwritten
    p3 = nd.matmul(dX, dW)
    e3 = nd.multiply(p3, 3.0)
    p3 += 1.0
    _close(e3.get(), 3 * (X.astype(np.float64) @ W), 2e-6)
    _close(p3.get(), X.astype(np.float64) @ W + 1, 2e-6)

    # the whole cfg4 sweep through the tape: forward in the GEMM epilogue, backward from the mask
    hip, oracle = engines
    s0 = dict(nd.FUSION_STATS)
    st, step = workloads.make_cfg4(hip, batch=512, d_in=128, d_out=1024)
    out = step()
    assert nd.FUSION_STATS["gemm_epilogue"] - s0["gemm_epilogue"] == 1
    Xh, Wh, bh = st["X"].as_numpy(), st["W"].as_numpy(), st["b"].as_numpy()
    zz = Xh.astype(np.float64) @ Wh + bh
    mm = np.asarray(zz > 0)
    _close(out["out"].as_numpy(), np.where(mm, zz, 0).sum(), 1e-5)
    gW, gb = st["W"].grad.as_numpy(), st["b"].grad.as_numpy()
    # gradients for the DEVICE's own mask would need the mask; near-zero pre-activations are ~1e-5 of the entries at most
    expW, expb = Xh.astype(np.float64).T @ mm, mm.sum(axis=0, dtype=np.float64)
    assert np.abs(gb - expb).max() <= 2 and np.abs(gW - expW).max() <= 2e-5 * np.abs(expW).max() + 8
    if os.environ.get("MDHIP_JIT") != "0":
        assert nd.FUSION_STATS["vm_eval_reduce_cols"] - s0["vm_eval_reduce_cols"] == 1   # g*mask + column sum: one pass over the MASK


def test_gemm_epilogue_cpu(lazy_nd, on_gpu, engines): _gemm_epilogue(lazy_nd, on_gpu, False, engines)
@gpu
def test_gemm_epilogue_gpu(lazy_nd, on_gpu, engines): _gemm_epilogue(lazy_nd, on_gpu, True, engines)
