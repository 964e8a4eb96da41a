import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rng = np.random.default_rng(0)
for shp, ax in (((8, 62500, 8), 1), ((8, 20000, 8), 1), ((20000, 8), 0), ((4, 16384, 16), 1), ((62500, 64), 0), ((2, 40000, 2), 1)):
    a = rng.integers(0, 1000, shp).astype(np.float32)
    d = nd.asarray(a)
    assert np.array_equal(nd.argmax(d, axis=ax).get(), np.argmax(a, axis=ax))
    lib.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); nd.argmax(d, axis=ax); lib.sync(); ts.append(time.perf_counter() - t0)
    print(shp, ax, f"{min(ts)*1e3:.3f} ms", flush=True)
