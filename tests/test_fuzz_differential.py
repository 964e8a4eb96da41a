"""Differential fuzzing against NumPy (tests/fuzz_device.py): random shapes, dtypes, strided /
transposed / broadcast views over the elementwise, reduction, arg-reduction, indexing, scatter and
matmul entry points; ints / bools / indices bit for bit, floats within ulp-scaled bounds."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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


import fuzz_tape  # noqa: E402


def test_fuzz_tape_cpu(lib, on_gpu):
    """Random forward+backward graphs (second order every fourth case): device table vs NumPy oracle
    table under the same tape."""
    if on_gpu:
        pytest.skip("other twin")
    assert fuzz_tape.main(250, 301) == 0


@pytest.mark.gpu
def test_fuzz_tape_gpu(lib, on_gpu):
    assert on_gpu
    assert fuzz_tape.main(800, 302) == 0


@pytest.mark.gpu
def test_fuzz_tape_lazy_gpu(lib, on_gpu):
    assert on_gpu
    from minidiff_amd import ndarray as nd
    prev = nd.set_lazy(True)
    try:
        assert fuzz_tape.main(500, 303) == 0
    finally:
        nd.set_lazy(prev)


def test_fuzz_lazy_cpu(lib, on_gpu):
    if on_gpu:
        pytest.skip("other twin")
    from minidiff_amd import ndarray as nd
    prev = nd.set_lazy(True)
    try:
        assert fuzz_tape.main(250, 304) == 0      # (found: a[idx] on a pending operand had no block to index)
        assert fuzz_device.main(400, 305, False) == 0
    finally:
        nd.set_lazy(prev)


import fuzz_axes  # noqa: E402


def test_fuzz_axes_cpu(lib, on_gpu):
    """Three- / four-axis iteration spaces (broadcast, sliced, axis-swapped operands), eager and fused (tests/fuzz_axes.py)."""
    if on_gpu:
        pytest.skip("other twin")
    assert fuzz_axes.main(160, 401) == 0


@pytest.mark.gpu
def test_fuzz_axes_gpu(lib, on_gpu):
    assert on_gpu
    assert fuzz_axes.main(1200, 402) == 0
