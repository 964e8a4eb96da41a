#!/usr/bin/env python3
"""A handful of launches of one GEMM configuration (for rocprofv3 --pmc passes): MDHIP_GEMM_CFG selects the tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
A = nd.asarray(rng.standard_normal((n, n), dtype=np.float32)); B = nd.asarray(rng.standard_normal((n, n), dtype=np.float32))
At = nd.asarray(np.ascontiguousarray(np.asarray(A).T))
for _ in range(6):
    nd.matmul(At.T, B)   # TN: both operands staged with vector LDS stores
lib.sync()
