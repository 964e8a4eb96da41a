#!/usr/bin/env python3
"""Host-side cost of one sweep (enqueue only, no device wait) against its device time: is a workload launch-bound?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd, workloads
from minidiff_amd.tape import hip_engine
lib = _capi.load()
md = hip_engine()
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
