"""Differential fuzzing against NumPy (scripts/fuzz_device.py): random shapes, dtypes, strided /
transposed / broadcast views over the elementwise, reduction, arg-reduction, indexing, scatter and
matmul entry points; ints / bools / indices bit for bit, floats within ulp-scaled bounds."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import fuzz_device  # noqa: E402


def test_fuzz_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    assert fuzz_device.main(500, 101, False) == 0
    assert fuzz_device.main(60, 102, True) == 0


@pytest.mark.gpu
def test_fuzz_gpu(lib, on_gpu):
    assert on_gpu
    assert fuzz_device.main(2500, 201, False) == 0
    assert fuzz_device.main(400, 202, True) == 0


@pytest.mark.gpu
def test_fuzz_lazy_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    prev = nd.set_lazy(True)
    try:
        assert fuzz_device.main(1500, 203, False) == 0
    finally:
        nd.set_lazy(prev)
