"""Why does bench.py's event bracket of cfg4's mask product read ~35 us when rocprofv3 says 29.7? (lab script)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rng = np.random.default_rng(0)
mask = nd.asarray(rng.random((8192, 4096)) > 0.5)
g = nd.asarray(np.float32(1.0))                       # the seed: 0-d
gb = nd.broadcast_to(g, (8192, 4096))
def ev():
    e = C.c_void_p(); lib.event_create(C.byref(e)); return e
def timed(fn, attach, n=20):
    out = []
    ms = C.c_float()
    for _ in range(n):
        e0, e1 = ev(), ev()
        if attach:
            lib.event_attach_next(e0, e1)
            r = fn()
            p = C.c_int(0); lib.event_attach_cancel(C.byref(p))
            assert not p.value, "nothing attachable was launched"
        else:
            lib.event_record(e0); r = fn(); lib.event_record(e1)
        lib.sync()
        lib.event_elapsed_ms(e0, e1, C.byref(ms)); out.append(ms.value * 1e3)
        del r
    out.sort()
    return f"min {out[0]:.1f} med {out[len(out)//2]:.1f} max {out[-1]:.1f} us"
for name, fn in (("seed(0-d) * mask", lambda: nd.multiply(g, mask)), ("broadcast view * mask", lambda: nd.multiply(gb, mask)), ("mask * seed", lambda: nd.multiply(mask, g)),
                 ("f32 array * mask", None)):
    if fn is None:
        x = nd.asarray(rng.standard_normal((8192, 4096)).astype(np.float32))
        fn = lambda: nd.multiply(x, mask)
    print(f"{name:24s} attached: {timed(fn, True)}   markers: {timed(fn, False)}")
# back-to-back, as in a sweep: a column sum right before (its tail still draining?)
z = nd.asarray(rng.standard_normal((8192, 4096)).astype(np.float32))
def seq():
    s = nd.sum(z)           # loss sum (k_reduce_all)
    return nd.multiply(gb, mask)
print("after a 134-MB reduction   markers around the pair:", timed(seq, False))
def seq_attached():
    s = nd.sum(z)
    e0, e1 = ev(), ev()
    lib.event_attach_next(e0, e1)
    r = nd.multiply(gb, mask)
    p = C.c_int(0); lib.event_attach_cancel(C.byref(p))
    lib.sync()
    ms = C.c_float(); lib.event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value * 1e3
v = sorted(seq_attached() for _ in range(20))
print(f"mask product right after a reduction, attached: min {v[0]:.1f} med {v[10]:.1f} max {v[-1]:.1f} us")
