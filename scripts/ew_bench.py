#!/usr/bin/env python3
"""GB/s of the streaming kernels at cfg3 / cfg4 shapes (HIP events, algorithmic bytes)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402


def main():
    lib = _capi.use_library(os.environ['MDHIP_AB_LIB']) if os.environ.get('MDHIP_AB_LIB') else _capi.load()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    rng = np.random.default_rng(0)
    x = nd.asarray(rng.standard_normal(N, dtype=np.float32))
    y = nd.asarray(rng.standard_normal(N, dtype=np.float32))
    R, Cc = 8192, 4096
    z = nd.asarray(rng.standard_normal((R, Cc), dtype=np.float32))
    bias = nd.asarray(rng.standard_normal((Cc,), dtype=np.float32))
    mask = nd.greater(z, 0)
    seed = nd.broadcast_to(nd.asarray(np.float32(1.0)), (R, Cc))
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    ms = C.c_float()
    cases = [
        ("copy 8N", lambda: nd.copy(x), 8 * N),
        ("square 8N", lambda: nd.power(x, 2), 8 * N),
        ("sin 8N", lambda: nd.sin(x), 8 * N),
        ("cos 8N", lambda: nd.cos(x), 8 * N),
        ("mul 12N", lambda: nd.multiply(x, y), 12 * N),
        ("mul scalar 8N", lambda: nd.multiply(x, 2.0), 8 * N),
        ("fill 4N", lambda: nd.ones((N,), dtype=np.float32), 4 * N),
        ("sum 4N", lambda: nd.sum(x), 4 * N),
        ("bias add", lambda: nd.add(z, bias), 8 * R * Cc),
        ("greater ->bool", lambda: nd.greater(z, 0), 5 * R * Cc),
        ("where(mask,z,0)", lambda: nd.where(mask, z, 0), 9 * R * Cc),
        ("seed*mask", lambda: nd.multiply(seed, mask), 5 * R * Cc),
        ("colsum (bias grad)", lambda: nd.sum(z, axis=(0,)), 4 * R * Cc),
        ("rowsum", lambda: nd.sum(z, axis=1), 4 * R * Cc),
        ("sum all 2d", lambda: nd.sum(z), 4 * R * Cc),
    ]
    for name, fn, nbytes in cases:
        for _ in range(3):
            fn()
        best = 1e9
        for _ in range(5):
            lib.event_record(e0)
            for _ in range(5):
                fn()
            lib.event_record(e1)
            lib.event_elapsed_ms(e0, e1, C.byref(ms))
            best = min(best, ms.value / 5)
        print("%-22s %8.3f ms  %7.1f GB/s" % (name, best, nbytes / best / 1e6))


if __name__ == "__main__":
    main()
