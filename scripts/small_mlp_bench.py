#!/usr/bin/env python3
"""A LAUNCH-bound sweep: the two-layer MLP of tests/test_training_loop.py at sizes where every kernel is microseconds. Sweeps per second
with the tape run in Python every step (eager kernels, then lazy fusion) against the same sweep captured once and replayed by
SweepCache (graph.py: the hipGraph counterpart of the reference's reuse_graph, minidiff/caching.py). usage: small_mlp_bench.py [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402
from minidiff_amd.graph import SweepCache  # noqa: E402
from minidiff_amd.hip_backend import HipBackendTable  # noqa: E402
from minidiff_amd.tape import build_engine  # noqa: E402

lib = _capi.load()
md = build_engine(HipBackendTable, "dev")
for batch, d_in, d_h, d_out in ((96, 24, 32, 5), (int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 256, 256, 16)):
    rng = np.random.default_rng(21)
    X = md.Tensor(rng.standard_normal((batch, d_in)).astype(np.float32))
    Y = md.Tensor(rng.standard_normal((batch, d_out)).astype(np.float32))
    P = [md.Tensor((rng.standard_normal((d_in, d_h)) * 0.2).astype(np.float32), allow_grad=True), md.Tensor(np.zeros(d_h, np.float32), allow_grad=True),
         md.Tensor((rng.standard_normal((d_h, d_out)) * 0.2).astype(np.float32), allow_grad=True), md.Tensor(np.zeros(d_out, np.float32), allow_grad=True)]

    def sweep():
        for p in P:
            p.grad = None
        loss = md.mean((md.tanh(X @ P[0] + P[1]) @ P[2] + P[3] - Y) ** 2)
        loss.backward()
        return {"loss": loss, "grads": [p.grad for p in P]}

    def rate(run, n=400):
        for _ in range(20):
            run()
        lib.sync()
        t = time.perf_counter()
        for _ in range(n):
            run()
        lib.sync()
        return n / (time.perf_counter() - t)

    print(f"MLP {batch} x {d_in} -> {d_h} -> {d_out}, forward + backward (float32):")
    nd.set_lazy(False)
    print(f"  tape in Python, eager kernels      {rate(sweep):9.0f} sweeps/s")
    nd.set_lazy(True)
    print(f"  tape in Python, lazy fusion        {rate(sweep):9.0f} sweeps/s")
    nd.set_lazy(False)
    with SweepCache(md, validate_every=0) as cache:
        r = rate(lambda: cache.run(sweep))
        print(f"  one captured hipGraph, replayed    {r:9.0f} sweeps/s   {cache.stats}")
