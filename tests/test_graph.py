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
