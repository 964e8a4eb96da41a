#!/usr/bin/env python3
"""Launch time of one GEMM layout against K at fixed M x N (HIP events over back-to-back launches): the slope is the
cost of a k-tile, the intercept what a launch pays besides its main loop (prologue, C write-out, launch tail).
usage: gemm_ksweep.py [M N] ; MDHIP_GEMM_CFG / MDHIP_GEMM_GLDS select the kernel."""
import ctypes as C
import os
os.environ.setdefault("MDHIP_EXPERIMENTS", "1")   # MDHIP_GEMM_CFG / MDHIP_GEMM_GLDS select the kernel: read only behind this gate
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402

lib = _capi.load()
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
rng = np.random.default_rng(0)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.event_create(C.byref(e0)); lib.event_create(C.byref(e1))
ms = C.c_float()
KS = (1024, 2048, 4096, 8192)
for layout in ("NN", "NT", "TN"):
    ts = []
    for K in KS:
        A = rng.standard_normal((M, K), dtype=np.float32)
        B = rng.standard_normal((K, N), dtype=np.float32)
        if layout == "NN":
            a, b = nd.asarray(A), nd.asarray(B)
        elif layout == "NT":
            a, b = nd.asarray(A), nd.asarray(np.ascontiguousarray(B.T)).T
        else:
            a, b = nd.asarray(np.ascontiguousarray(A.T)).T, nd.asarray(B)
        for _ in range(20):
            nd.matmul(a, b)
        best = 1e9
        for _ in range(3):
            lib.event_record(e0)
            for _ in range(20):
                nd.matmul(a, b)
            lib.event_record(e1)
            lib.event_elapsed_ms(e0, e1, C.byref(ms))
            best = min(best, ms.value / 20 * 1e3)
        ts.append(best)
    slope = (ts[3] - ts[2]) / (KS[3] - KS[2])
    icpt = ts[2] - slope * KS[2]
    ideal = 2.0 * M * N / 157.3e12 * 1e6   # us per unit of K at the MFMA peak
    print(f"{layout}: " + "  ".join(f"K={k}: {t:7.1f} us ({2.0*M*N*k/t/1e6:6.1f} TF)" for k, t in zip(KS, ts)))
    print(f"     slope {slope*1e3:.2f} ns per k (peak {ideal*1e3:.2f}: {ideal/slope*100:.1f} % in the main loop), intercept {icpt:.1f} us")
