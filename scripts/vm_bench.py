#!/usr/bin/env python3
"""Cost of the expression interpreter per program shape (HIP events, N = 1e8 f32)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402


def main():
    lib = _capi.load()
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    rng = np.random.default_rng(0)
    x = nd.asarray(rng.standard_normal(N, dtype=np.float32))
    y = nd.asarray(rng.standard_normal(N, dtype=np.float32))
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    ms = C.c_float()
    nd.set_lazy(True)
    seed = nd.broadcast_to(nd.asarray(np.float32(1.0)), (N,))

    def xgrad():
        s = nd.sin(x); m = nd.multiply(s, y)
        g = nd.multiply(nd.multiply(nd.multiply(seed, 2), nd.power(m, 1)), y)
        return nd.multiply(g, nd.cos(x))

    cases = [
        ("copy-ish x*1 (8N)", lambda: nd.multiply(x, 1.0), 8 * N),
        ("x*y (12N)", lambda: nd.multiply(x, y), 12 * N),
        ("x*y+x*y*2 (12N, 7 instr)", lambda: nd.add(nd.multiply(x, y), nd.multiply(nd.multiply(x, y), 2.0)), 12 * N),
        ("sin(x) (8N)", lambda: nd.sin(x), 8 * N),
        ("sin(x)*y (12N)", lambda: nd.multiply(nd.sin(x), y), 12 * N),
        ("sin(x)*cos(x) (8N)", lambda: nd.multiply(nd.sin(x), nd.cos(x)), 8 * N),
        ("x*y*x*y*x*y*x*y (12N, 7 ops)", lambda: nd.multiply(nd.multiply(nd.multiply(nd.multiply(nd.multiply(nd.multiply(nd.multiply(x, y), x), y), x), y), x), y), 12 * N),
        ("(x*y)**1 (12N)", lambda: nd.power(nd.multiply(x, y), 1), 12 * N),
        ("(x*y)**2 (12N)", lambda: nd.power(nd.multiply(x, y), 2), 12 * N),
        ("seed*2*(x*y) (12N)", lambda: nd.multiply(nd.multiply(seed, 2), nd.multiply(x, y)), 12 * N),
        ("cfg3 x.grad (12N)", xgrad, 12 * N),
        ("cfg3 fwd sum((sin x*y)^2) (8N)", lambda: nd.sum(nd.power(nd.multiply(nd.sin(x), y), 2)), 8 * N),
    ]
    for name, fn, nbytes in cases:
        for _ in range(2):
            nd.materialize(fn())
        best = 1e9
        for _ in range(4):
            lib.event_record(e0)
            for _ in range(3):
                nd.materialize(fn())
            lib.event_record(e1)
            lib.event_elapsed_ms(e0, e1, C.byref(ms))
            best = min(best, ms.value / 3)
        print("%-36s %8.3f ms  %7.1f GB/s" % (name, best, nbytes / best / 1e6))


if __name__ == "__main__":
    main()
