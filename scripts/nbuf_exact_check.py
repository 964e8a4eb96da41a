import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from minidiff_amd import _capi, ndarray as nd
lib = _capi.load()
rng = np.random.default_rng(5)
for (M,K,N) in ((128,64,128),(256,96,384),(2048,2048,2048),(1024,4096,4096),(128,32,128),(384,160,256)):
    a = rng.integers(-3,4,(M,K)).astype(np.float32); b = rng.integers(-3,4,(K,N)).astype(np.float32)
    bt = np.ascontiguousarray(b.T)
    ref = a.astype(np.float64) @ b
    for cfg in ("7","5","0","6"):
        lib.debug_set_option(b"gemm_cfg", int(cfg))
        for nb in ("3","2"):
            lib.debug_set_option(b"gemm_nbuf", int(nb))
            at = np.ascontiguousarray(a.T)
            for tag, x, y in (("NN", nd.asarray(a), nd.asarray(b)), ("NT", nd.asarray(a), nd.asarray(bt).T), ("TN", nd.asarray(at).T, nd.asarray(b))):
                got = nd.matmul(x, y).get()
                assert np.array_equal(got, ref), (M,K,N,cfg,nb,tag)
print("nbuf exact ok")
