#!/usr/bin/env python3
"""np.add.at / a[idx] = v at ELEMENT granularity on the device: sorted-by-destination (default) against the bid / apply rounds
(experiment option scatter_sorted = 0), by multiplicity of the destinations. HIP events around the call; host time where the rounds
synchronise. usage: scatter_bench.py [N]"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("MDHIP_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402

lib = _capi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(0)
vals = nd.asarray(rng.standard_normal(N).astype(np.float32))
print(f"{N} float32 contributions, np.add.at on a 1-D destination; wall time per call incl. the final sync (median of 5)")
for name, dests in (("all distinct (a permutation)", N), ("~4 per destination", N // 4), ("~1000 per destination", max(N // 1000, 1)), ("10 destinations (a histogram)", 10), ("one destination", 1)):
    idx_h = rng.permutation(N) if dests == N else rng.integers(0, dests, N)
    idx = nd.asarray(idx_h)
    out = {}
    for mode in (1, 0):
        if mode == 0 and N // dests > 20000:
            out[mode] = float("nan")     # (one round and one read-back per multiplicity level: minutes)
            continue
        lib.debug_set_option(b"scatter_sorted", mode)
        ts = []
        for _ in range(5):
            a = nd.zeros((max(dests, 16),), np.float32)
            lib.sync()
            t = time.perf_counter()
            nd.index_add(a, idx, vals)
            lib.sync()
            ts.append((time.perf_counter() - t) * 1e3)
        out[mode] = sorted(ts)[2]
    lib.debug_set_option(b"scatter_sorted", 1)
    ref = np.zeros(max(dests, 16), np.float32)
    np.add.at(ref, idx_h, vals.get())
    a = nd.zeros((max(dests, 16),), np.float32)
    nd.index_add(a, idx, vals)
    assert np.array_equal(a.get(), ref)
    print(f"  {name:32s} sorted {out[1]:9.3f} ms   rounds {out[0]:9.3f} ms", flush=True)
