"""What a GEMM launch pays besides its k-loop: the kernel's own duration (events attached to the dispatch) at small K, for output
sizes with different C traffic and tile counts. (lab script)"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402
lib = _capi.load()
rng = np.random.default_rng(0)
ms = C.c_float()
def ev():
    e = C.c_void_p(); lib.event_create(C.byref(e)); return e
def own_time(a, b, out=None, n=30):
    ts = []
    for _ in range(5):
        nd.matmul(a, b, out=out)
    lib.sync()
    for _ in range(n):
        e0, e1 = ev(), ev()
        lib.event_attach_next(e0, e1)
        r = nd.matmul(a, b, out=out)
        p = C.c_int(0); lib.event_attach_cancel(C.byref(p))
        lib.sync()
        lib.event_elapsed_ms(e0, e1, C.byref(ms)); ts.append(ms.value * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]
print("M x N        tiles  C MiB | K: min / median us of the kernel itself")
for M, N in ((1024, 1024), (2048, 2048), (1024, 4096), (4096, 4096), (2048, 4096)):
    row = []
    for K in (32, 64, 128, 256, 512, 1024, 2048):
        a = nd.asarray(rng.standard_normal((M, K), dtype=np.float32))
        b = nd.asarray(rng.standard_normal((K, N), dtype=np.float32))
        out = nd.zeros((M, N), np.float32)
        mn, md_ = own_time(a, b, out)
        row.append((K, mn, md_))
    (k1, t1, _), (k2, t2, _) = row[-2], row[-1]
    slope = (t2 - t1) / (k2 - k1)
    print(f"{M}x{N:<6d} C {M*N*4/2**20:5.1f} MiB | " + "  ".join(f"{k}: {mn:.1f}/{md_:.1f}" for k, mn, md_ in row) + f" | slope {slope*1e3:.1f} ns/k -> {2.0*M*N/slope/1e6:.1f} TF, intercept {t2 - slope*k2:.1f} us")
