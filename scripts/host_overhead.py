#!/usr/bin/env python3
"""Host-side cost of one sweep (enqueue only, no device wait) against its device time: is a workload launch-bound?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd, workloads
from minidiff_amd.tape import hip_engine
lib = _capi.load()
md = hip_engine()
if sys.argv[1] == "calls":
    # host cost of single eager backend calls (8 elements: the device side is nothing), C route (csrc/fastpath.c) vs the Python
    # implementation behind it; run again with MDHIP_FASTPATH=0 for the all-Python path of round 2 (Python block owner too)
    import numpy as np
    a = nd.asarray(np.ones(8, dtype=np.float32)); b = nd.asarray(np.ones(8, dtype=np.float32))
    m = nd.asarray(np.ones((8, 8), dtype=np.float32)); c = nd.asarray(np.ones((8, 8), dtype=np.bool_))

    def t(f, n=200, bursts=60):
        """median over short bursts, each started on an idle stream: a long loop of tiny kernels fills the queue and then
        measures the DEVICE's rate (~2 us per tiny kernel), not the host's"""
        for _ in range(2000):
            f()
        out = []
        for _ in range(bursts):
            lib.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            out.append((time.perf_counter() - t0) / n * 1e6)
        lib.sync()
        return sorted(out)[len(out) // 2]
    print("fast path:", "on" if nd._fp is not None else "off (MDHIP_FASTPATH=0)")
    r = nd.DeviceArray._new((8,), a.dtype)
    da, db, dr = a.desc(), b.desc(), r.desc()
    if nd._fp is not None:
        floor = sorted((lib.sync(), nd._fp.time_binary(_capi.B_ADD, a, b, 200))[1] / 200 * 1e6 for _ in range(60))[30]
        print("%-44s%6.2f us   <- mdhip_binary from C, descriptors ready: dispatch + hipLaunchKernel" % ("C-ABI floor", floor))
    print("%-44s%6.2f us" % ("lib.binary through ctypes, descriptors ready", t(lambda: lib.binary(_capi.B_ADD, da, db, dr, _capi.F32))))
    for name, f in (("add(a, b)", lambda: nd.add(a, b)), ("multiply(a, 2.0)", lambda: nd.multiply(a, 2.0)), ("sin(a)", lambda: nd.sin(a)),
                    ("a + b (dunder)", lambda: a + b), ("sum(a)", lambda: nd.sum(a)), ("sum(m, axis=0)", lambda: nd.sum(m, axis=0)),
                    ("matmul(m, m)  [8 x 8]", lambda: nd.matmul(m, m)), ("where(c, m, 0)", lambda: nd.where(c, m, 0)), ("multiply(m, c)  [float x bool]", lambda: nd.multiply(m, c)),
                    ("DeviceArray._new", lambda: nd.DeviceArray._new((8,), a.dtype))):
        print("%-44s%6.2f us" % (name, t(f)))
    if nd._fp is not None:
        for name, f in (("add(a, b)  [Python implementation]", lambda: nd.add.__wrapped__(a, b)), ("sin(a)  [Python implementation]", lambda: nd.sin.__wrapped__(a)),
                        ("matmul(m, m)  [Python implementation]", lambda: nd.matmul.__wrapped__(m, m)), ("sum(m, axis=0)  [Python implementation]", lambda: nd.sum.__wrapped__(m, axis=0))):
            print("%-44s%6.2f us" % (name, t(f)))
    sys.exit(0)
if sys.argv[1] == "tiny_mlp":
    # an 8 x 8 x 8 MLP forward + backward (matmul, bias add, greater, where, sum; backward: mask product, two matmuls, column sum): every
    # kernel is trivial, the sweep is host dispatch
    state, step = workloads.make_cfg4(md, batch=8, d_in=8, d_out=8)
    for _ in range(200):
        step()
    out = []
    for _ in range(30):
        lib.sync()
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        out.append((time.perf_counter() - t0) / 100 * 1e6)
    lib.sync()
    print("tiny MLP sweep, fast path %s: %.1f us of host dispatch per sweep (median of 30 bursts of 100)" % ("on" if nd._fp is not None else "off", sorted(out)[15]))
    sys.exit(0)
wl, lazy = sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "lazy"
nd.set_lazy(lazy)
state, step = workloads.MAKERS[wl](md)
for _ in range(10):
    step()
lib.sync()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
lib.sync()
t2 = time.perf_counter()
print(f"{wl} {'lazy' if lazy else 'eager'}: host enqueue {1e3*(t1-t0)/n:.3f} ms/sweep, wall {1e3*(t2-t0)/n:.3f} ms/sweep")
if len(sys.argv) > 3:
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        step()
    pr.disable(); lib.sync()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
