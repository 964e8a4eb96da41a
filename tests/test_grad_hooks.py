"""Gradient-ready hooks of the tape (minidiff_amd/tape.py `register_grad_ready_hook`): what the
data-parallel harness uses to start a collective while backward() is still running."""
import numpy as np


def test_hooked_input_is_served_first_and_fires_before_the_other_vjp(engines):
    hip, _ = engines
    rng = np.random.default_rng(0)
    A = hip.Tensor(rng.standard_normal((6, 5)).astype(np.float32), allow_grad=True)
    B = hip.Tensor(rng.standard_normal((5, 7)).astype(np.float32), allow_grad=True)
    seen = []

    def hook(t):
        assert t is B and t.grad is not None
        seen.append(A.grad is None)  # A's vjp (listed first in the op) has not been evaluated yet

    hip.register_grad_ready_hook(B, hook)
    try:
        (A @ B).backward()
        assert seen == [True]
        gB = B.grad.as_numpy().copy()
        gA = A.grad.as_numpy().copy()
    finally:
        hip.remove_grad_ready_hook(B)
    A.grad = B.grad = None
    (A @ B).backward()
    assert seen == [True]
    np.testing.assert_array_equal(B.grad.as_numpy(), gB)
    np.testing.assert_array_equal(A.grad.as_numpy(), gA)


def test_hook_fires_once_after_the_last_contribution(engines):
    hip, ora = engines
    w = np.random.default_rng(1).standard_normal((4, 4)).astype(np.float32)
    x = np.random.default_rng(2).standard_normal((4, 4)).astype(np.float32)
    out = {}
    for name, md in (("hip", hip), ("ora", ora)):
        W = md.Tensor(w, allow_grad=True)
        X = md.Tensor(x, allow_grad=True)
        calls = []
        md.register_grad_ready_hook(W, lambda t: calls.append(t.grad.as_numpy().copy()))
        try:
            f = md.sum(W * W + (X @ W) * md.sin(W))  # W contributes through four inputs of three ops
            f.backward()
        finally:
            md.remove_grad_ready_hook(W)
        assert len(calls) == 1
        np.testing.assert_array_equal(calls[0], W.grad.as_numpy())  # it was final when the hook ran
        out[name] = (W.grad.as_numpy(), X.grad.as_numpy())
    np.testing.assert_allclose(out["hip"][0], out["ora"][0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["hip"][1], out["ora"][1], rtol=1e-5, atol=1e-6)


def test_no_hooks_no_bookkeeping(engines):
    hip, _ = engines
    assert hip.grad_ready_hooks == {}


class _DoublingComm:
    """Stand-in communicator of a 2-rank world whose other rank holds the same gradients: an all-reduce doubles the buffer.
    Records every collective (sizes), synchronous and 'asynchronous' (same thing here)."""
    world, rank = 2, 0

    def __init__(self):
        self.calls = []

    def allreduce_sum_(self, arr):
        self.calls.append(int(arr.size))
        arr *= 2

    allreduce_sum_async_ = allreduce_sum_

    def wait(self):
        self.calls.append("wait")


def _mlp(md, K=512, N=24, rows=16, extra=False):
    rng = np.random.default_rng(7)
    X = md.Tensor(rng.standard_normal((rows, K)).astype(np.float32))
    W = md.Tensor((rng.standard_normal((K, N)) / 16).astype(np.float32), allow_grad=True)
    b = md.Tensor(rng.standard_normal((N,)).astype(np.float32), allow_grad=True)
    c = md.Tensor(rng.standard_normal((N,)).astype(np.float32), allow_grad=True) if extra else None

    def step(reset=True):
        if reset:
            W.grad = b.grad = None
            if c is not None:
                c.grad = None
        z = X @ W + b
        md.sum(md.where(z > 0, z, 0)).backward()

    return X, W, b, c, step


def test_gradsync_bias_rides_on_the_last_panel_and_unused_parameter_raises(engines):
    """with the weight gradient sent in panels the REST of the bucket must not be forgotten. The bias
    (final before W in backward order, next to W in the bucket) rides on W's last panel; a registered parameter the sweep
    never touches makes the end-of-sweep call raise instead of silently leaving rank-local gradients."""
    from minidiff_amd import dp
    _, ora = engines
    X, W, b, c, step = _mlp(ora)
    step()
    ref = {"W": W.grad.as_numpy().copy(), "b": b.grad.as_numpy().copy()}
    comm = _DoublingComm()
    sync = dp.GradSync(ora, [W, b], comm, panels=2)
    try:
        step()
        assert comm.calls == [256 * 24, 256 * 24 + 24]          # two W panels, the second carrying the bias: 2 collectives, not 3
        sync()
        assert comm.calls[-1] == "wait" and sync.overlapped == 1
        np.testing.assert_allclose(W.grad.as_numpy(), 2 * ref["W"], rtol=1e-6)
        np.testing.assert_allclose(b.grad.as_numpy(), 2 * ref["b"], rtol=1e-6)
    finally:
        sync.close()
    # an unused third parameter: W (and b with it) went out in panels, c has no gradient -> the sweep's end raises
    X, W, b, c, step = _mlp(ora, extra=True)
    comm = _DoublingComm()
    sync = dp.GradSync(ora, [W, b, c], comm, panels=2)
    try:
        step()
        import pytest
        with pytest.raises(RuntimeError, match="no gradient"):
            sync()
        # .. and the object is usable again afterwards (its per-sweep state was reset): c gets a gradient by hand
        step()
        c.grad = ora.Tensor(np.ones(24, dtype=np.float32))
        sync()
        np.testing.assert_array_equal(c.grad.as_numpy(), np.full(24, 2, dtype=np.float32))
    finally:
        sync.close()
    # a parameter whose gradient exists but was never reported through a hook (overlap off) is reduced at the call
    X, W, b, c, step = _mlp(ora)
    comm = _DoublingComm()
    sync = dp.GradSync(ora, [W, b], comm, overlap=False)
    step()
    sync()
    assert comm.calls == [512 * 24 + 24]                         # panels = 1 semantics: the literal single all-reduce
    np.testing.assert_allclose(b.grad.as_numpy(), 2 * ref["b"], rtol=1e-6)


def test_gradsync_second_backward_before_the_join_is_refused(engines):
    """Two backward() calls before the GradSync object is called would accumulate rank-local values onto an already
    reduced bucket view: refused loudly."""
    import pytest
    from minidiff_amd import dp
    _, ora = engines
    X, W, b, c, step = _mlp(ora)
    sync = dp.GradSync(ora, [W, b], _DoublingComm(), panels=2)
    try:
        step()
        with pytest.raises(RuntimeError, match="GradSync"):
            step(reset=False)
    finally:
        sync.close()


def test_auto_panels_by_shape():
    from minidiff_amd.dp import auto_panels
    assert auto_panels(4096, 4096) == 4        # cfg4 / cfg2: 1024-row panels, 256 tiles of 128 x 128 each
    assert auto_panels(1024, 1024) == 1        # below one round of tiles: the single all-reduce
    assert auto_panels(8192, 8192) == 8        # capped
    assert auto_panels(16384, 256) == 1
