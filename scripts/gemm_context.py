#!/usr/bin/env python3
"""Does a GEMM's place in the cfg2 sweep change its duration? Kernel-attached events (mdhip_event_attach_next) around each product
of several orderings of the three 4096^3 products on the sweep's own buffers (A, B, G = ones, outputs re-allocated as in the tape)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402

lib = _capi.load()
n = 4096
rng = np.random.default_rng(0)
A = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
B = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
G = nd.ones((n, n), dtype=np.float32)


def timed(fn, store, tag):
    e0, e1 = C.c_void_p(), C.c_void_p()
    lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
    lib.event_attach_next(e0, e1)
    out = fn()
    store.append((tag, e0, e1))
    return out


ops = {
    "NN": lambda: nd.matmul(A, B),
    "NT": lambda: nd.matmul(G, B.T),
    "TN": lambda: nd.matmul(A.T, G),
    "fill": lambda: nd.ones((n, n), dtype=np.float32),
    "big": lambda: nd.ones((16384, 8192), dtype=np.float32),      # 512 MiB write: pushes everything out of the Infinity Cache
}
for order in (["NN", "fill", "NT", "TN"], ["NN", "NT", "TN"], ["NT", "TN", "NN"], ["TN", "NN", "NT"], ["NN", "NN", "NT", "NT", "TN", "TN"],
              ["big", "NN", "big", "NT", "big", "TN"]):
    for _ in range(5):
        for o in order:
            ops[o]()
    lib.sync()
    store = []
    for _ in range(20):
        for o in order:
            if o in ("fill", "big"):
                ops[o]()
            else:
                timed(ops[o], store, o)
    lib.sync()
    ms = C.c_float()
    acc = {}
    for i, (tag, e0, e1) in enumerate(store):
        lib.event_elapsed_ms(e0, e1, C.byref(ms))
        pos = [k for k in order if k not in ("fill", "big")]
        slot = i % len(pos)
        acc.setdefault((slot, tag), []).append(ms.value * 1e3)
    print(" -> ".join(order) + ":  " + "  ".join(f"{tag}[{slot}] {np.mean(v):7.1f} us" for (slot, tag), v in sorted(acc.items())))
