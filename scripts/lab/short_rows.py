"""Row reductions over MANY SHORT rows (wall time, GB/s of the bytes read). (lab script)"""
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
    ms = min(ts) * 1e3
    print(f"{name:46s} {ms:9.3f} ms  {nbytes / min(ts) / 1e9:8.1f} GB/s", flush=True)
for rows, cols in ((4_000_000, 4), (4_000_000, 16), (1_000_000, 64), (1_000_000, 268), (100_000, 1000), (4_000_000, 268)):
    for dt in (np.float32, np.bool_, np.int8, np.float64, np.int64):
        if rows * cols * np.dtype(dt).itemsize > 6e9:
            continue
        x = nd.asarray((rng.random((rows, cols)) > 0.999) if dt is np.bool_ else rng.integers(0, 9, (rows, cols)).astype(dt))
        nb = rows * cols * np.dtype(dt).itemsize
        nm = f"{rows}x{cols} {np.dtype(dt).name}"
        if dt is np.bool_:
            t(f"any axis=-1 {nm}", lambda: nd.any(x, axis=-1), nb)
            t(f"all axis=-1 {nm}", lambda: nd.all(x, axis=-1), nb)
        else:
            t(f"sum axis=-1 {nm}", lambda: nd.sum(x, axis=-1), nb)
            t(f"max axis=-1 {nm}", lambda: nd.max(x, axis=-1), nb)
        t(f"argmax axis=-1 {nm}", lambda: nd.argmax(x, axis=-1), nb)
        del x
