"""A few column sums of one tall shape for rocprofv3 (lab script). usage: colsum_one.py ROWS COLS"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rows, cols = int(sys.argv[1]), int(sys.argv[2])
x = nd.asarray(np.random.default_rng(0).standard_normal((rows, cols)).astype(np.float32))
for _ in range(10):
    nd.sum(x, axis=0); nd.max(x, axis=0)
lib.sync()
