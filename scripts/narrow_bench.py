#!/usr/bin/env python3
"""Streaming rate of the native storage-only-dtype kernels (csrc/narrow.hip): one launch per call, each operand read once in its
own type. HIP-event timing, algorithmic bytes = operand bytes + result bytes. `NARROW_ONLY=name` runs one case a few times (for
rocprofv3 --pmc passes: profiles/r4_pmc_narrow.csv).   usage: narrow_bench.py [N]   (default 2**30 elements)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402

lib = _capi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
rng = np.random.default_rng(0)


def make(dt):
    # (host side in chunks: 2**30 random int64 at once would be 8 GiB)
    out = nd.empty((N,), dt) if hasattr(nd, "empty") else nd.DeviceArray.empty((N,), np.dtype(dt))
    step = 1 << 26
    for i in range(0, N, step):
        n = min(step, N - i)
        h = (rng.integers(1, 100, n).astype(dt) if np.dtype(dt).kind != "f" else rng.standard_normal(n).astype(dt))
        out[i:i + n] = nd.asarray(h)
    return out


cases = [("int8 * int8", np.int8, "multiply", 3), ("uint8 + uint8", np.uint8, "add", 3), ("int8 < int8 -> bool", np.int8, "less", 3),
         ("float16 + float16", np.float16, "add", 6), ("int16 * int16", np.int16, "multiply", 6), ("uint32 max uint32", np.uint32, "maximum", 12),
         ("int8 * 3 (scalar)", np.int8, "multiply_scalar", 2), ("-int8", np.int8, "negative", 2), ("sum(int8) -> int64", np.int8, "sum", 1),
         ("max(float16)", np.float16, "max", 2)]
only = os.environ.get("NARROW_ONLY")
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
ms = C.c_float()
cache = {}
for name, dt, op, bpe in cases:
    if only and only != name:
        continue
    if dt not in cache:
        cache.clear()
        cache[dt] = (make(dt), make(dt))
    a, b = cache[dt]
    if op == "multiply_scalar":
        fn = lambda: nd.multiply(a, 3)          # noqa: E731
    elif op in ("negative",):
        fn = lambda: nd.negative(a)             # noqa: E731
    elif op in ("sum", "max"):
        fn = lambda op=op: getattr(nd, op)(a)   # noqa: E731
    else:
        fn = lambda op=op: getattr(nd, op)(a, b)   # noqa: E731
    # parity on a slice first
    with np.errstate(all="ignore"):
        if op in ("sum", "max"):
            pass
        else:
            ha, hb = a[:100003].get(), b[:100003].get()
            ref = {"multiply_scalar": lambda: ha * 3, "negative": lambda: -ha}.get(op, lambda: getattr(np, op)(ha, hb))()
            got = fn()[:100003].get()
            assert np.array_equal(got, ref) if ref.dtype.kind != "f" else np.allclose(got, ref, rtol=2e-3), name
    for _ in range(2):
        fn()
    best = 1e9
    reps = 3 if only else 5
    for _ in range(reps):
        lib.event_record(e0)
        for _ in range(3):
            r = fn()
        lib.event_record(e1)
        lib.event_elapsed_ms(e0, e1, C.byref(ms))
        best = min(best, ms.value / 3)
    nbytes = bpe * N
    print("%-26s %9.3f ms  %7.1f GB/s  = %5.1f %% of 8 TB/s  (%d bytes per element)" % (name, best, nbytes / best / 1e6, nbytes / best / 1e6 / 80.0, bpe), flush=True)
