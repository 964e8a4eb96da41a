#!/usr/bin/env python3
"""Skinny products (matrix x vector, vector x matrix, a few columns / rows): HBM-bound — the big operand is read once.
Prints time and the share of 8 TB/s that one read of the big operand in that time amounts to."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402


def main():
    lib = _capi.load()
    rng = np.random.default_rng(0)
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    ms = C.c_float()
    R = int(os.environ.get("SKINNY_N", "8192"))
    A = nd.asarray(rng.standard_normal((R, R), dtype=np.float32))
    for nc in (1, 2, 4, 8):
        v = nd.asarray(rng.standard_normal((R, nc), dtype=np.float32))
        vt = nd.asarray(rng.standard_normal((nc, R), dtype=np.float32))
        cases = [("A @ v", lambda: nd.matmul(A, v), lambda: A.get().astype(np.float64) @ v.get()),
                 ("A.T @ v", lambda: nd.matmul(A.T, v), lambda: A.get().T.astype(np.float64) @ v.get()),
                 ("vt @ A", lambda: nd.matmul(vt, A), lambda: vt.get().astype(np.float64) @ A.get()),
                 ("vt @ A.T", lambda: nd.matmul(vt, A.T), lambda: vt.get().astype(np.float64) @ A.get().T)]
        if nc == 1:
            v1 = nd.asarray(v.get()[:, 0].copy())
            cases.append(("A @ v (1-D)", lambda: nd.matmul(A, v1), lambda: A.get().astype(np.float64) @ v1.get()))
        for name, f, ref in cases:
            out = f()
            r = ref()
            err = np.abs(out.get() - r).max() / np.abs(r).max()
            for _ in range(3):
                f()
            ts = []
            for _ in range(7):
                lib.event_record(e0)
                for _ in range(5):
                    f()
                lib.event_record(e1)
                lib.event_elapsed_ms(e0, e1, C.byref(ms))
                ts.append(ms.value / 5 * 1e3)
            t = sorted(ts)[len(ts) // 2]
            print("nc=%d %-12s %8.1f us  %5.1f %% of 8 TB/s   rel err %.1e" % (nc, name, t, R * R * 4 / (t * 1e-6) / 8e12 * 100, err), flush=True)


if __name__ == "__main__":
    main()
