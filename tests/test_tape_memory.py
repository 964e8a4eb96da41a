"""Intermediates must be released by refcount alone (no cyclic-GC dependence):
on the device every leaked temporary is an allocator block that cannot be
recycled, which turns the next sweep's allocations into hipMallocs."""
import gc

import numpy as np


def test_sweep_releases_intermediates_without_gc(lib, engines):
    import ctypes as C
    dev, _ = engines
    from minidiff_amd import workloads
    st, step = workloads.make_cfg3(dev, n=4096)
    s = (C.c_int64 * 4)()
    gc.collect()
    gc.disable()
    try:
        step()
        lib.mem_stats(s)
        base = s[0]
        for _ in range(5):
            step()
        lib.mem_stats(s)
        assert s[0] == base, (base, s[0])
    finally:
        gc.enable()
