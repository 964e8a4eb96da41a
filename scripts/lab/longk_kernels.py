"""A few launches of the long-k product kernels for `rocprofv3 --kernel-trace --stats` (lab script)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rng = np.random.default_rng(0)
K = 1 << 22
x, y = nd.asarray(rng.standard_normal(K).astype(np.float32)), nd.asarray(rng.standard_normal(K).astype(np.float32))
A, B = nd.asarray(rng.standard_normal((8, K >> 2)).astype(np.float32)), nd.asarray(rng.standard_normal((K >> 2, 8)).astype(np.float32))
A64, B64 = nd.asarray(rng.standard_normal((64, K >> 2))), nd.asarray(rng.standard_normal((K >> 2, 64)))
for _ in range(20):
    nd.dot(x, y); nd.matmul(A, B); nd.matmul(A, B[:, 0]); nd.matmul(A64, B64)
lib.sync()
