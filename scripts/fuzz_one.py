#!/usr/bin/env python3
"""Re-run ONE case of tests/fuzz_device.py (seed, index[, --big]) and print what differed (for a reduce case: value, NumPy's value, the
float64-exact reference by log-sum for products)."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_device as fz  # noqa: E402
from minidiff_amd import ndarray as nd  # noqa: E402

seed, idx = int(sys.argv[1]), int(sys.argv[2])
big = "--big" in sys.argv
calls = []
for name in ("prod", "sum", "mean"):
    orig = getattr(nd, name)

    def wrap(d, *a, _o=orig, _n=name, **k):
        r = _o(d, *a, **k)
        calls.append((_n, d, a, k, r))
        return r
    setattr(nd, name, wrap)
try:
    fz.one_case(np.random.default_rng([seed, idx]), big)
    print("case passed")
except AssertionError as e:
    print("FAIL:", e)
for name, d, a, k, r in calls:
    h = np.asarray(d)
    exp = getattr(np, name)(h, *a, **k)
    got = np.asarray(r)
    print(name, h.shape, h.dtype, k, "got", got.ravel()[:3], "numpy", np.asarray(exp).ravel()[:3], "rel diff", np.abs(got - exp).max() / (np.abs(exp).max() + 1e-300))
    if name == "prod" and h.dtype.kind == "f" and got.size == 1:
        ref = math.exp(math.fsum(np.log(np.abs(h.astype(np.float64))).ravel()))
        print("   |exact| %.17g   ours off by %.3e   numpy off by %.3e (relative)" % (ref, abs(abs(float(got.ravel()[0])) - ref) / ref, abs(abs(float(np.asarray(exp).ravel()[0])) - ref) / ref))
