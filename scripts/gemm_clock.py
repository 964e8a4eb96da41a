#!/usr/bin/env python3
"""In-kernel clock and main-loop time of the fp32 GEMM under SUSTAINED load: MDHIP_GEMM_STAMP=1 makes libmdhip stamp
s_memtime / s_memrealtime around every block's main loop into a persistent buffer (no sync behind the launches) and print the
last launch's figures at exit.   usage: gemm_clock.py N LAYOUT [launches]"""
import os, sys
os.environ["MDHIP_EXPERIMENTS"] = "1"   # the library reads experiment variables only behind this gate
os.environ["MDHIP_GEMM_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
n, layout = int(sys.argv[1]), sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
rng = np.random.default_rng(0)
A = nd.asarray(rng.standard_normal((n, n), dtype=np.float32)); B = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
a, b = {"NN": lambda: (A, B), "NT": lambda: (A, nd.asarray(np.ascontiguousarray(np.asarray(B).T)).T),
        "TN": lambda: (nd.asarray(np.ascontiguousarray(np.asarray(A).T)).T, B)}[layout]()
e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
for _ in range(reps - 50):
    nd.matmul(a, b)
lib.event_record(e0)
for _ in range(50):
    nd.matmul(a, b)
lib.event_record(e1)
lib.sync()
lib.event_elapsed_ms(e0, e1, C.byref(ms))
print(f"{layout} {n}^3: {ms.value / 50 * 1e3:.1f} us per launch over the last 50 of {reps} back-to-back launches = {2.0 * n ** 3 / (ms.value / 50 * 1e-3) / 1e12:.1f} TFLOP/s", file=sys.stderr)
