#!/usr/bin/env python3
"""Row scatters for `rocprofv3 --kernel-trace --stats`: np.add.at with unique / duplicated row indices, a[perm] = w."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minidiff_amd import _capi, ndarray as nd  # noqa: E402

lib = _capi.load()
rng = np.random.default_rng(0)
R, C = 8192, 4096
z = nd.asarray(rng.standard_normal((R, C), dtype=np.float32))
w = nd.asarray(rng.standard_normal((R, C), dtype=np.float32))
perm = nd.asarray(rng.permutation(R))
idx64 = nd.asarray(rng.integers(0, 64, (R,)))
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for _ in range(10):
    if which in ("all", "unique"):
        nd.index_add(z, (perm,), w)
    if which in ("all", "dups"):
        nd.index_add(z, (idx64,), w)
    if which in ("all", "set"):
        z[perm] = w
lib.sync()
