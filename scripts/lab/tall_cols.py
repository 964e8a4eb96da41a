"""Column reductions (axis 0) of TALL matrices with few to a thousand columns (wall time, GB/s). (lab script)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rng = np.random.default_rng(0)
def t(name, fn, nbytes):
    fn(); lib.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); lib.sync(); ts.append(time.perf_counter() - t0)
    print(f"{name:44s} {min(ts)*1e3:9.3f} ms  {nbytes / min(ts) / 1e9:8.1f} GB/s", flush=True)
for rows, cols in ((4_000_000, 16), (4_000_000, 17), (1_000_000, 64), (1_000_000, 268), (1_000_000, 267), (300_000, 1000), (100_000, 4096)):
    for dt in (np.float32, np.float64, np.bool_):
        x = nd.asarray((rng.random((rows, cols)) > 0.5) if dt is np.bool_ else rng.standard_normal((rows, cols)).astype(dt))
        nb = rows * cols * np.dtype(dt).itemsize
        nm = f"{rows}x{cols} {np.dtype(dt).name}"
        if dt is np.bool_:
            t(f"any axis=0 {nm}", lambda: nd.any(x, axis=0), nb)
        else:
            t(f"sum axis=0 {nm}", lambda: nd.sum(x, axis=0), nb)
            t(f"max axis=0 {nm}", lambda: nd.max(x, axis=0), nb)
            t(f"argmax axis=0 {nm}", lambda: nd.argmax(x, axis=0), nb)
        del x
