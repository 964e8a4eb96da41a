#!/usr/bin/env python3
"""The round-3 kernels that are not on a BASELINE config, 20 calls each, for `rocprofv3 --kernel-trace --stats` (kernel times without
the host between calls): thin products, std, arg-reductions, sums over a last / middle axis of a 3-D array, row gathers, TT and
peeled products."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402


def main():
    lib = _capi.load()
    rng = np.random.default_rng(0)
    z = nd.asarray(rng.standard_normal((8192, 4096), dtype=np.float32))
    t3 = nd.asarray(rng.standard_normal((64, 512, 1024), dtype=np.float32))
    A = nd.asarray(rng.standard_normal((8192, 8192), dtype=np.float32))
    v = nd.asarray(rng.standard_normal((8192, 1), dtype=np.float32))
    idx = nd.asarray(rng.integers(0, 8192, (8192,)))
    P = nd.asarray(rng.standard_normal((4096, 4097), dtype=np.float32))     # x.T of this: M = 4097 (odd leading dimension)
    Q = nd.asarray(rng.standard_normal((4096, 4100), dtype=np.float32))
    S = nd.asarray(rng.standard_normal((4096, 4096), dtype=np.float32))
    calls = [lambda: nd.argmax(z, axis=0), lambda: nd.argmax(z, axis=1), lambda: nd.std(z, axis=1), lambda: nd.std(z, axis=0),
             lambda: nd.sum(t3, axis=2), lambda: nd.sum(t3, axis=1), lambda: nd.matmul(A, v), lambda: nd.matmul(A.T, v), lambda: z[idx],
             lambda: nd.matmul(S.T, S.T), lambda: nd.matmul(P.T, Q)]
    only = os.environ.get("MISC_ONLY")          # e.g. "0,10": indices into the list above
    if only:
        calls = [calls[int(i)] for i in only.split(",")]
    for f in calls:
        for _ in range(20):
            f()
        lib.sync()


if __name__ == "__main__":
    main()
