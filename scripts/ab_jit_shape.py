#!/usr/bin/env python3
"""A/B of the generated streaming kernels' shape on cfg3's fused gradient pass (k_fused_eval2: reads x, y, writes x.grad, y.grad =
16N bytes): vector groups per lane and trip (option jit_u) x blocks per CU (jit_blocks), interleaved rounds in ONE process."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd, workloads  # noqa: E402
from minidiff_amd.tape import hip_engine  # noqa: E402

lib = _capi.load()
md = hip_engine()
nd.set_lazy(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
state, step = workloads.make_cfg3(md, n=n)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
ms = C.c_float()
res = {}
combos = [(u, b) for u in (2, 3) for b in (1, 2, 3, 8)] + [(1, 2), (1, 3)]
for rnd in range(4):
    for u, b in combos:
        lib.debug_set_option(b"jit_u", u)
        lib.debug_set_option(b"jit_blocks", b)
        step(); step()
        lib.sync()
        lib.event_record(e0)
        for _ in range(5):
            step()
        lib.event_record(e1)
        lib.event_elapsed_ms(e0, e1, C.byref(ms))
        res.setdefault((u, b), []).append(ms.value / 5)
for (u, b), v in res.items():
    v.sort()
    med = v[len(v) // 2]
    print(f"jit_u {u} jit_blocks {b}:  sweep med {med*1e3:7.1f} us  min {v[0]*1e3:7.1f} us   ({24.0 * n / (med * 1e-3) / 1e12:5.2f} TB/s of 24N bytes)")
