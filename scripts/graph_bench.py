"""Latency of a SMALL sweep (the README graph, second order included; reference README.md:13-36)
run eagerly vs replayed from a captured hipGraph. Small graphs are bound by host dispatch
(Python tape + ctypes + launch), which a replay removes.   python scripts/graph_bench.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from minidiff_amd import _capi, ndarray as nd
from minidiff_amd.tape import hip_engine
from minidiff_amd.graph import CapturedSweep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _capi.load()
md = hip_engine()
rng = np.random.default_rng(0)
x = md.Tensor(rng.standard_normal((2, n // 2)).astype(np.float32), allow_grad=True)
y = md.Tensor(rng.standard_normal((2, n // 2)).astype(np.float32), allow_grad=True)
mat = md.backend._materialize


def step():
    x.grad = None
    y.grad = None
    f = 2 * y * md.sin(x) - x ** 2
    f.backward(allow_higher_order=True)
    x.grad.backward()
    mat(x.grad._data)
    mat(y.grad._data)
    return x, y


def timeit(fn, reps):
    fn(); lib.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    lib.sync()
    return (time.perf_counter() - t0) / reps * 1e6


for lazy in (False, True):
    nd.set_lazy(lazy)
    for _ in range(5):
        step()
    eager_us = timeit(step, 200)
    ref = (x.grad.as_numpy().copy(), y.grad.as_numpy().copy())
    sweep = CapturedSweep(step)
    graph_us = timeit(sweep.replay, 2000)
    ok = np.array_equal(ref[0], x.grad.as_numpy()) and np.array_equal(ref[1], y.grad.as_numpy())
    print(f"n={n} lazy={lazy}: dispatch per sweep {eager_us:8.1f} us   graph replay {graph_us:7.1f} us   x{eager_us / graph_us:.1f}   identical={ok}", flush=True)
    sweep.close()
nd.set_lazy(False)
