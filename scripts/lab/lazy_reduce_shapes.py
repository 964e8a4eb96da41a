"""Fused elementwise + reduction in lazy mode over awkward shapes (wall time; eager twin beside it). (lab script)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rng = np.random.default_rng(0)
def t(fn):
    fn(); lib.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = fn(); nd.materialize(r); lib.sync(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
for shp in ((2000, 2000), (1_000_000, 268), (268, 1_000_000), (4_000_000, 16), (16, 4_000_000), (200, 200, 200), (100_000, 10, 10)):
    x = nd.asarray(rng.standard_normal(shp).astype(np.float32)); y = nd.asarray(rng.standard_normal(shp).astype(np.float32))
    for ax in (None,) + tuple(range(len(shp))) + ((0, 2),) * (len(shp) == 3):
        f = lambda: nd.sum(nd.multiply(nd.sin(x), y), axis=ax)
        nd.set_lazy(False); e = t(f)
        nd.set_lazy(True); l = t(f)
        nd.set_lazy(False)
        flag = "   <-- lazy slower" if l > 1.5 * e and l > 0.5 else ""
        print(f"sum(sin(x)*y, axis={ax}) {str(shp):22s} eager {e:8.3f} ms   lazy {l:8.3f} ms{flag}", flush=True)
    del x, y
