#!/usr/bin/env python3
"""(lab: needs the un-kept option gemm_nt_c patched into gemm.hip — GemmArgs::nt_c and a nontemporal store in the two epilogues)
A/B of the experiment option gemm_nt_c (C written with nontemporal stores by the direct-to-LDS GEMM kernels): interleaved rounds
of back-to-back launches per shape and layout, HIP events, median of the rounds; then the chained product C2 = (A @ B) @ B, where
the consumer re-reads C. usage: ab_gemm_nt_c.py [MxKxN ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402

lib = _capi.load()
opt = lambda v: lib.debug_set_option(b"gemm_nt_c", int(v))   # noqa: E731
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
ms = C.c_float()
rng = np.random.default_rng(0)
shapes = [(4096, 4096, 4096), (2048, 2048, 2048), (1024, 4096, 4096), (4096, 1024, 4096), (8192, 4096, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]


def timed(fn, reps):
    lib.event_record(e0)
    for _ in range(reps):
        fn()
    lib.event_record(e1)
    lib.sync()
    lib.event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / reps * 1e3


for (M, K, N) in shapes:
    A = nd.asarray(rng.standard_normal((M, K), dtype=np.float32))
    B = nd.asarray(rng.standard_normal((K, N), dtype=np.float32))
    At = nd.asarray(np.ascontiguousarray(A.get().T))
    Bt = nd.asarray(np.ascontiguousarray(B.get().T))
    out = nd.zeros((M, N), np.float32)
    cases = {"NN": lambda: nd.matmul(A, B, out=out), "NT": lambda: nd.matmul(A, Bt.T, out=out), "TN": lambda: nd.matmul(At.T, B, out=out)}
    reps = 10 if M * N * K >= 2 ** 34 else 30
    for fn in cases.values():
        for _ in range(6):
            fn()
    res = {(k, v): [] for k in cases for v in (0, 1)}
    for _ in range(7):
        for v in (0, 1):
            opt(v)
            for k, fn in cases.items():
                res[(k, v)].append(timed(fn, reps))
    opt(0)
    flop = 2.0 * M * N * K
    line = []
    for k in cases:
        a, b = sorted(res[(k, 0)])[3], sorted(res[(k, 1)])[3]
        line.append(f"{k}: plain {a:7.1f} us ({flop / a / 1e6:6.1f} TF)  nt {b:7.1f} us ({flop / b / 1e6:6.1f} TF)  {100 * (a / b - 1):+5.2f} %")
    print(f"{M}x{K}x{N}  " + " | ".join(line), flush=True)
# the consumer reads C right away (cfg5's chain): does the nontemporal C cost the next kernel its L2 hits?
for n in (2048, 4096):
    A = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
    B = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
    c1, c2 = nd.zeros((n, n), np.float32), nd.zeros((n, n), np.float32)
    def chain():
        nd.matmul(A, B, out=c1)
        nd.matmul(c1, B, out=c2)
    for _ in range(6):
        chain()
    r = {0: [], 1: []}
    for _ in range(7):
        for v in (0, 1):
            opt(v)
            r[v].append(timed(chain, 10))
    opt(0)
    a, b = sorted(r[0])[3], sorted(r[1])[3]
    print(f"chain (A@B)@B n={n}: plain {a:7.1f} us  nt {b:7.1f} us  {100 * (a / b - 1):+5.2f} %", flush=True)
