#!/usr/bin/env python3
"""The finite-difference gradient checker at a non-toy size: `compute_grads` (tests/fdcheck.py, the counterpart of
the reference's minidiff/utils.py:104-197) for C = A @ B with (64, 64) @ (64, 64) float64 — n = 4096 elements per input, 2 x 4096
perturbed evaluations per input, 16384 in all — on the device with vmap as a Python row loop, with vmap replaying ONE captured hipGraph
per row (minidiff_amd/hip_backend.py), and on the NumPy engine. Prints seconds and the agreement of the two gradients."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fdcheck  # noqa: E402
from minidiff_amd import hip_backend, ndarray as nd  # noqa: E402
from minidiff_amd.tape import build_engine, hip_engine  # noqa: E402
from oracle.numpy_table import NumpyOracleTable  # noqa: E402  (the checker's CPU leg: test infrastructure)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
ah, bh = rng.standard_normal((n, n)), rng.standard_normal((n, n))


def run(md, label):
    a, b = md.Tensor(ah, allow_grad=True), md.Tensor(bh, allow_grad=True)
    f = lambda x, y: md.sum(md.matmul(x, y) ** 2) / 2      # noqa: E731  (half squared sum: tests/test_ops.py:25-62's loss)
    t0 = time.perf_counter()
    manual, auto = fdcheck.compute_grads(md, a, b, func=f, h=1e-6)
    m = [np.asarray(g.as_numpy()) for g in manual]
    u = [np.asarray(g.as_numpy()) for g in auto]
    dt = time.perf_counter() - t0
    err = max(float(np.abs(x - y).max() / np.abs(y).max()) for x, y in zip(m, u))
    print(f"{label:46s} {dt:8.3f} s   max |fd - autodiff| / max|autodiff| = {err:.2e}   ({4 * n * n} evaluations of f)", flush=True)
    return m


hip = hip_engine()
hip_backend.VMAP_REPLAY = False
run(hip, "device, vmap = Python row loop (warm-up)") if n <= 32 else None
g_loop = run(hip, "device, vmap = Python row loop")
hip_backend.VMAP_REPLAY = True
g_rep = run(hip, "device, vmap = hipGraph replay per row")
g_rep = run(hip, "device, vmap = hipGraph replay per row (again)")
assert all(np.array_equal(x, y) for x, y in zip(g_loop, g_rep)), "replayed rows must equal the looped rows bit for bit"
ref = build_engine(NumpyOracleTable, "oracle")
g_np = run(ref, "NumPy engine (host)")
print("device vs NumPy engine, finite differences: max rel", max(float(np.abs(x - y).max() / np.abs(y).max()) for x, y in zip(g_rep, g_np)))
