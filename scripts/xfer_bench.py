#!/usr/bin/env python3
"""Host<->device transfer rates through the C-ABI (mdhip_h2d / mdhip_d2h), and the PCIe-inclusive
rate of the headline sweep (cfg2 with A and B uploaded every sweep and both gradients read back)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402
from minidiff_amd.tape import hip_engine  # noqa: E402

lib = _capi.load()
for mib in (1, 16, 64, 256):
    h = np.random.default_rng(0).standard_normal(mib * (1 << 18), dtype=np.float32)
    d = nd.asarray(h)
    lib.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        d = nd.asarray(h)
    lib.sync()
    up = 5 * h.nbytes / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    for _ in range(5):
        back = np.asarray(d)
    down = 5 * h.nbytes / (time.perf_counter() - t0) / 1e9
    print(f"{mib:4d} MiB  H2D {up:6.1f} GB/s   D2H {down:6.1f} GB/s", flush=True)

md = hip_engine()
n = 4096
rng = np.random.default_rng(2)
a = rng.standard_normal((n, n), dtype=np.float32)
b = rng.standard_normal((n, n), dtype=np.float32)


def sweep():
    A = md.Tensor(a, allow_grad=True)
    B = md.Tensor(b, allow_grad=True)
    C = A @ B
    C.backward()
    return A.grad.as_numpy(), B.grad.as_numpy()


sweep()
t0 = time.perf_counter()
for _ in range(5):
    sweep()
dt = (time.perf_counter() - t0) / 5
print(f"cfg2 with 2 x 64 MiB uploaded and 2 x 64 MiB read back per sweep: {dt * 1e3:.2f} ms/sweep = {1 / dt:.1f} passes/s (PCIe-inclusive; resident-input figure is bench.py's value)")
