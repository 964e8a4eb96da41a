"""The device backend (HIP table behind the C-ABI) against the reference's
golden vectors: forward, loss, gradients, exception types and the backend-call
traces (which also pin that views stay views — contiguity flags are part of a
trace entry). CPU twin runs on the test double; the gpu-marked twin is the
parity test proper. Bounds: ints/bools/indices bit-exact, fp32 1e-5, fp64 1e-12
(norm-wise, north_star)."""
import numpy as np
import pytest

import golden_util as gu
from test_oracle_golden import CONFIGS, run_config

gpu = pytest.mark.gpu


def _engine(lib):
    from minidiff_amd.hip_backend import HipBackendTable
    from minidiff_amd.tape import build_engine
    return build_engine(HipBackendTable, "dev")


@pytest.fixture
def lazy_mode(request):
    """`lazy` twins: the same golden cases with lazy fusion on (minidiff_amd/lazy.py: fused elementwise chains, deferred products,
    owed column sums) — so the driver's own `-m gpu` run covers the fused path on the reference's vectors, not only builder logs."""
    from minidiff_amd import ndarray as nd
    prev = nd.set_lazy(request.param == "lazy")
    yield request.param
    nd.set_lazy(prev)


MODES = pytest.mark.parametrize("lazy_mode", ["eager", "lazy"], indirect=True)


def _case(lib, on_gpu, want_gpu, key):
    if want_gpu:
        assert on_gpu and lib.target == "hip:gfx950"
    elif on_gpu:
        pytest.skip("GPU present: covered by the gpu-marked twin")
    gu.run_case_on(_engine(lib), key, exact=False)


@MODES
@pytest.mark.parametrize("key", gu.case_keys())
def test_device_matches_reference_cpu(lib, on_gpu, key, lazy_mode): _case(lib, on_gpu, False, key)


@gpu
@MODES
@pytest.mark.parametrize("key", gu.case_keys())
def test_device_matches_reference_gpu(lib, on_gpu, key, lazy_mode): _case(lib, on_gpu, True, key)


def _config(lib, on_gpu, want_gpu, name):
    if want_gpu:
        assert on_gpu and lib.target == "hip:gfx950"
    elif on_gpu:
        pytest.skip("GPU present: covered by the gpu-marked twin")
    from minidiff_amd.hip_backend import HipBackendTable
    res, trace = run_config(HipBackendTable, name)
    g = gu.golden()
    for k, v in res.items():
        exp = g["cfg"][f"{name}/{k}"]
        assert v.dtype == exp.dtype and v.shape == exp.shape, (name, k, v.dtype, exp.dtype)
        if exp.dtype.kind in "biu":
            assert np.array_equal(v, exp), (name, k)
        else:
            tol = 1e-5 if exp.dtype == np.float32 else 1e-12
            assert gu.rel_err(v, exp) <= tol, (name, k, gu.rel_err(v, exp))
    # host->device uploads (and NumPy's scalar re-wrap of full reductions, tensor.py:102-103)
    # appear as tensor_constructor calls on one side only; every other entry must match
    exp_trace = [t for t in g["info"]["traces"][name] if t[0] != "tensor_constructor"]
    trace = [t for t in trace if t[0] != "tensor_constructor"]
    assert [t[0] for t in trace] == [t[0] for t in exp_trace], name
    for i, (a, b) in enumerate(zip(trace, exp_trace)):
        assert a == b, (name, i, a, b)


@MODES
@pytest.mark.parametrize("name", CONFIGS)
def test_device_config_cpu(lib, on_gpu, name, lazy_mode): _config(lib, on_gpu, False, name)


@gpu
@MODES
@pytest.mark.parametrize("name", CONFIGS)
def test_device_config_gpu(lib, on_gpu, name, lazy_mode): _config(lib, on_gpu, True, name)
