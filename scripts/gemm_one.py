#!/usr/bin/env python3
"""A handful of launches of one GEMM shape and layout (for rocprofv3 --pmc passes): MDHIP_GEMM_CFG selects the tile.
usage: gemm_one.py M K N [NN|NT|TN]"""
import os, sys
os.environ.setdefault("MDHIP_EXPERIMENTS", "1")   # MDHIP_GEMM_CFG / MDHIP_GEMM_GLDS select the kernel: read only behind this gate
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 4096)
layout = sys.argv[4] if len(sys.argv) > 4 else "TN"
rng = np.random.default_rng(0)
A = nd.asarray(rng.standard_normal((M, K), dtype=np.float32))
B = nd.asarray(rng.standard_normal((K, N), dtype=np.float32))
if layout == "NN":
    a, b = A, B
elif layout == "NT":
    a, b = A, nd.asarray(np.ascontiguousarray(np.asarray(B).T)).T
else:
    a, b = nd.asarray(np.ascontiguousarray(np.asarray(A).T)).T, B
for _ in range(6):
    nd.matmul(a, b)
lib.sync()
